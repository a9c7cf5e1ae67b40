#!/bin/bash
# Compile the HiOp-side adapters of the `hip-native` back-end against the REFERENCE's own headers and link them against
# libhiopamd.so.  Proves (a) every pure virtual of hiopVector / hiopMatrixDense / hiopMatrixSparse / hiopLinSolverSymDense
# and of the user interface hiopInterfaceMDS (MdsEx1HipNative) is overridden with the reference's exact signature
# (`override` everywhere, and one object of each class is instantiated),
# (b) every C-ABI symbol the adapters use exists in the library.
#
# Nothing of the reference is copied into the repo: the two headers cmake would generate (hiop_defs.hpp from
# src/Interface/hiop_defs.hpp.in, FortranCInterface.hpp) are generated into a temporary directory by this script and thrown
# away.  Runs on the build container only (needs /root/reference); the GPU box never needs it.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(dirname "$HERE")"
REF="${HIOP_REFERENCE:-/root/reference}"
[ -d "$REF/src" ] || { echo "check_adapters: reference tree not found at $REF (set HIOP_REFERENCE)"; exit 2; }
[ -f "$REPO/hiop_amd/lib/libhiopamd.so" ] || { echo "check_adapters: build libhiopamd.so first (python -c 'import __graft_entry__ as g; g.build()')"; exit 2; }
TMP="$(mktemp -d)"
trap '[ -n "${KEEP_TMP:-}" ] && echo "kept $TMP" || rm -rf "$TMP"' EXIT

# ---- the two cmake-generated headers (configuration: HIOP_USE_GPU + HIOP_USE_HIP, no MPI, no RAJA, no MAGMA) ----
sed -e 's/^#cmakedefine HIOP_USE_GPU$/#define HIOP_USE_GPU/' \
    -e 's/^#cmakedefine HIOP_USE_HIP$/#define HIOP_USE_HIP/' \
    -e 's/^#cmakedefine \(.*\)$/\/* #undef \1 *\//' \
    -e 's/@PROJECT_VERSION@/1.1.1/; s/@PROJECT_VERSION_MAJOR@/1/; s/@PROJECT_VERSION_MINOR@/1/; s/@PROJECT_VERSION_PATCH@/1/' \
    -e 's/@HIOP_RELEASE_DATE@/adapter-check/' \
    "$REF/src/Interface/hiop_defs.hpp.in" > "$TMP/hiop_defs.hpp"
cat > "$TMP/FortranCInterface.hpp" <<'EOF'
#pragma once
#define FC_GLOBAL(name, NAME) name##_
#define FC_GLOBAL_(name, NAME) name##_
EOF

INC=(-I"$TMP" -I"$REPO/include" -I"$HERE")
for d in Interface LinAlg Optimization Utils ExecBackends; do INC+=(-I"$REF/src/$d"); done
CXX="${CXX:-g++}"
FLAGS=(-std=c++14 -fPIC -O1 -Wall -Wextra -Wno-unused-parameter -Woverloaded-virtual -Werror=overloaded-virtual)

SRCS=(hiopVectorHipNative.cpp hiopVectorIntHipNative.cpp hiopMatrixDenseHipNative.cpp hiopMatrixSparseTripletHipNative.cpp hiopLinSolverSymDenseHipNative.cpp
      hiopLinSolverSymSparseHipNative.cpp
      MdsEx1HipNative.cpp DenseConsEx2HipNative.cpp LinAlgFactoryHipNative.cpp)
OBJS=()
for s in "${SRCS[@]}"; do
  o="$TMP/${s%.cpp}.o"
  echo "  CXX $s"
  "$CXX" "${FLAGS[@]}" "${INC[@]}" -c "$HERE/$s" -o "$o"
  OBJS+=("$o")
done
echo "  LD  libhiopamd_adapters.so"
"$CXX" -shared -o "$TMP/libhiopamd_adapters.so" "${OBJS[@]}" -L"$REPO/hiop_amd/lib" -lhiopamd -Wl,-rpath,"$REPO/hiop_amd/lib" -Wl,--no-undefined \
  -Wl,--unresolved-symbols=report-all 2> "$TMP/ld.err" || true

# ---- undefined symbols: anything that is neither ours (libhiopamd), libc/libstdc++, nor a reference-side non-virtual helper
# the HiOp build itself provides (hiopLinSolver base ctor/dtor, hiopVectorPar/Int accessors, logger) is a hole in the adapter.
ALLOWED='hiop::hiopLinSolver|hiop::hiopLinSolverSymDense|hiop::hiopVectorPar|hiop::hiopVectorInt|hiop::hiopMatrixDenseRowMajor|hiop::hiopLogger|hiop::hiopNlpFormulation|hiop::hiopOptions|hiop::LinearAlgebraFactory|typeinfo for hiop::|vtable for hiop::hiop(LinSolver|Vector|Matrix)[A-Za-z]*$'
UNDEF_OURS=$(grep -o "undefined reference to \`[^']*'" "$TMP/ld.err" | sed "s/undefined reference to \`//; s/'$//" | sort -u | grep -E '^hiopamd_' || true)
UNDEF_OTHER=$(grep -o "undefined reference to \`[^']*'" "$TMP/ld.err" | sed "s/undefined reference to \`//; s/'$//" | sort -u | grep -vE '^hiopamd_' | grep -vE "$ALLOWED" || true)
if [ -n "$UNDEF_OURS" ]; then echo "MISSING C-ABI symbols in libhiopamd.so:"; echo "$UNDEF_OURS"; exit 1; fi
if [ -n "$UNDEF_OTHER" ]; then echo "unexpected undefined symbols:"; echo "$UNDEF_OTHER"; exit 1; fi

# ---- unimplemented virtuals: a translation unit that instantiates one object of every adapter class only compiles when
# no pure virtual is left (g++ names each missing one in the error).
cat > "$TMP/instantiate.cpp" <<'EOF'
#include "hiopVectorHipNative.hpp"
#include "hiopVectorIntHipNative.hpp"
#include "hiopMatrixDenseHipNative.hpp"
#include "hiopMatrixSparseTripletHipNative.hpp"
#include "hiopLinSolverSymDenseHipNative.hpp"
#include "hiopLinSolverSymSparseHipNative.hpp"
#include "MdsEx1HipNative.hpp"
#include "DenseConsEx2HipNative.hpp"
using namespace hiop;
void* instantiate_all(hiopNlpFormulation* nlp)
{
  auto* v = new hiopVectorHipNative(8);
  auto* vi = new hiopVectorIntHipNative(8);
  auto* M = new hiopMatrixDenseHipNative(4, 8);
  auto* S = new hiopMatrixSparseTripletHipNative(4, 8, 6);
  auto* Y = new hiopMatrixSymSparseTripletHipNative(8, 6);
  auto* L = new hiopLinSolverSymDenseHipNative(8, nlp);
  auto* LS = new hiopLinSolverSymSparseHipNative(8, 6, nlp);   // (compile / link check: never executed)
  auto* LS2 = new hiopLinSolverSymSparseHipNative(Y, nlp);
  hiopInterfaceMDS* E = new MdsEx1HipNative(40, 12);   // the user-problem side: hiopInterfaceMDS on device pointers
  hiopInterfaceDenseConstraints* E2 = new DenseConsEx2HipNative(1000);
  static void* all[] = {v, vi, M, S, Y, L, LS, LS2, E, E2};
  return all;
}
EOF
if ! "$CXX" "${FLAGS[@]}" "${INC[@]}" -c "$TMP/instantiate.cpp" -o "$TMP/instantiate.o" 2> "$TMP/inst.err"; then
  echo "UNIMPLEMENTED VIRTUALS:"; grep -E "pure virtual|virtual .* = 0|because the following" -A2 "$TMP/inst.err" | head -80; exit 1
fi
# the same translation units must also compile in the reference's HIOP_DEEPCHECKS configuration (extra pure virtuals there)
for s in "${SRCS[@]}" ; do
  "$CXX" -DHIOP_DEEPCHECKS "${FLAGS[@]}" "${INC[@]}" -c "$HERE/$s" -o "$TMP/deep_${s%.cpp}.o"
done
"$CXX" -DHIOP_DEEPCHECKS "${FLAGS[@]}" "${INC[@]}" -c "$TMP/instantiate.cpp" -o "$TMP/deep_instantiate.o"

count_virtuals() { tr '\n' ' ' < "$1" | grep -oE "virtual [^;{}]*=\s*0\s*;" | wc -l; }
echo "pure virtuals in the reference headers: hiopVector $(count_virtuals "$REF/src/LinAlg/hiopVector.hpp"), hiopMatrix $(count_virtuals "$REF/src/LinAlg/hiopMatrix.hpp"), hiopMatrixDense $(count_virtuals "$REF/src/LinAlg/hiopMatrixDense.hpp") (+ $(grep -c 'not implemented in base class' "$REF/src/LinAlg/hiopMatrixDense.hpp") assert(false) bodies), hiopMatrixSparse $(count_virtuals "$REF/src/LinAlg/hiopMatrixSparse.hpp"), hiopLinSolver $(count_virtuals "$REF/src/LinAlg/hiopLinSolver.hpp")"
echo "overrides in the adapters: $(cat "$HERE"/*HipNative.hpp | grep -c ' override')"
echo "C-ABI symbols referenced by the adapters: $(nm -u "${OBJS[@]}" | grep ' U hiopamd_' | sort -u | wc -l) distinct (all resolved by libhiopamd.so)"
echo "methods that stop loudly instead of forwarding: $(grep -c 'hiopamd_not_in_path("' "$HERE"/*.cpp | awk -F: '{s+=$2} END {print s}')"
echo "unimplemented virtuals: 0"
echo "check_adapters: OK"
