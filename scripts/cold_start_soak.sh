#!/bin/bash
# Fresh-process soak of the C interface (tests/c/mds_c_interface.c): every solve is a NEW process — first launch of every code object,
# allocations nobody touched before — the shape of the driver's failed round-5 gate.
#   scripts/cold_start_soak.sh <runs per case> <out file> [case ...]      case = mode:ns:nd   (default: the list below)
# A run counts as failed when the program's exit code is not 0 (status != Solve_Success, or the reference driver's objective check).
set -u
R=${1:-100}; OUT=${2:-gpurun_out/cold_start_soak.txt}; shift 2 || true
CASES=${*:-"host:400:100 device:400:100 host:401:100 host:40:12 device:40:12 host:1000:1044 device:1000:1045"}
cd "$(dirname "$0")/.."
LIBDIR=hiop_amd/lib${HIOPAMD_BUILD_VARIANT:+_$HIOPAMD_BUILD_VARIANT}
EXE=/tmp/mds_c_interface_soak
gcc -std=c11 -O1 -Iinclude tests/c/mds_c_interface.c -o $EXE -L$LIBDIR -lhiopamd -lm -Wl,-rpath,$PWD/$LIBDIR || exit 1
mkdir -p "$(dirname "$OUT")"
echo "# cold-start soak: $R fresh processes per case, library $LIBDIR ($(sha256sum $LIBDIR/libhiopamd.so | cut -c1-16)), $(date -u +%FT%TZ)" >> $OUT
total=0; failed=0
for c in $CASES; do
  IFS=: read mode ns nd <<< "$c"
  f=0; t0=$(date +%s.%N)
  for i in $(seq 1 $R); do
    if ! timeout 120 $EXE $mode $ns $nd > /tmp/soak_out.txt 2> /tmp/soak_err.txt; then
      f=$((f+1))
      if [ $f -le 5 ]; then
        echo "--- FAILED run $i of case $c" >> $OUT
        grep -v "^ *[0-9]\+ " /tmp/soak_out.txt | tail -4 >> $OUT; sed -n 1,3p /tmp/soak_out.txt >> $OUT; tail -5 /tmp/soak_err.txt >> $OUT
      fi
    fi
  done
  t1=$(date +%s.%N)
  echo "case $c (N = $((nd + ns + 3))): $R runs, $f failed, $(python3 -c "print(f'{($t1-$t0)/$R:.3f}')") s per process; last: $(grep '^obj=' /tmp/soak_out.txt | tail -1)" | tee -a $OUT
  total=$((total+R)); failed=$((failed+f))
done
echo "TOTAL: $total fresh-process solves, $failed failed" | tee -a $OUT
[ $failed -eq 0 ]
