# Per-kernel register / scratch / occupancy / LDS table from hipcc's resource-usage remarks
# (runs anywhere hipcc does; no GPU needed).  bash scripts/kernel_resources.sh > profiles/...
cd "$(dirname "$0")/.."
for f in gnnflow_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c -x hip "$f" \
      -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1
done | python3 -c '
import re, subprocess, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: \s*(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.splitlines()
print("%-36s %6s %6s %8s %10s %8s" % ("kernel", "VGPRs", "SGPRs", "scratch", "occupancy", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("gf::(anonymous namespace)::", "").replace("void ", ""))
    print("%-36s %6s %6s %8s %10s %8s" % (n, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"),
                                          r.get("Occupancy"), r.get("LDS")))
'
