for r in 1 2 3 4; do
HIOPAMD_DEV_CUCOUNT=1 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-dense 2> /tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f it/s fact %.3f' % (d['value'], d['kkt_spans']['linsolv.tmFactTime']['ms_per_step']))"
grep "DEV cu" /tmp/err.txt | sort | uniq -c | sort -rn | head -4
done
