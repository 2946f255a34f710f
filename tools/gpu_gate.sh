#!/bin/bash
# usage: tools/gpu_gate.sh <outdir> <pytest args...> -- <command...>
# Runs the given pytest selection first; the command after "--" (benchmarks, profiles) only runs when the tests passed
# and the GPU runtime reported no memory fault.  Exits non-zero otherwise.
out=$1; shift
mkdir -p "$out"
targs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do targs+=("$1"); shift; done
shift
timeout -k 10 600 python -m pytest "${targs[@]}" -m gpu -x -q > "$out/pytest.log" 2>&1
rc=$?
tail -8 "$out/pytest.log"
if [ $rc -ne 0 ] || grep -q "Memory access fault" "$out/pytest.log"; then echo "GATE: tests failed (rc=$rc); skipping the rest"; exit 1; fi
"$@"
