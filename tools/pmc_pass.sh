#!/bin/bash
# One extra PMC pass over the bench workload: tools/pmc_pass.sh <outdir> <name> <counters...>
set -u
OUT=$(realpath -m "$1"); NAME=$2; shift 2
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc "$@" -d "$OUT/$NAME" -o "$NAME" --output-format csv -- \
    python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu > "$OUT/$NAME.log" 2>&1 || { echo "pass $NAME failed"; tail -3 "$OUT/$NAME.log"; exit 1; }
python3 "$ROOT/tools/summarize_pmc.py" "$OUT/$NAME" > "$OUT/$NAME.json" && echo "pass $NAME ok"
