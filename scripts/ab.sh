# Same-box A/B of the headline replay: bash scripts/ab.sh TAG "ENV_A" "ENV_B" [pairs] [extra bench args]
# Runs bench.py (replica leg only, no CPU baseline / config 3) alternately with the two
# environments (e.g. "GNNFLOW_LRU_FUSED=0" vs "") and prints us per step of every run.
# One parameterised script instead of one file per experiment (the round-4 table of
# experiments is in profiles/README.md).
TAG=$1; A=$2; B=$3; PAIRS=${4:-3}; shift 4 2>/dev/null
OUT=gpurun_out/ab_$TAG; mkdir -p $OUT
for i in $(seq 1 $PAIRS); do
  for side in A B; do
    if [ $side = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --no-hash-leg --no-config3 --no-cpu-baseline "$@" > $OUT/${side}_$i.json 2> $OUT/${side}_$i.err || exit 1
    python - "$OUT/${side}_$i.json" "$side" "$E" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("%s [%s] %.2f us/step  %.1f M edges/s  gather %.2f us frac %.3f  lru %s" % (
    sys.argv[2], sys.argv[3], d["ms_per_step"] * 1e3, d["value"] / 1e6,
    r.get("avg_launch_us", float("nan")), r.get("frac", float("nan")), d.get("roofline_lru")), flush=True)
PY
  done
done
