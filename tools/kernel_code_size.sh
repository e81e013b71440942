#!/bin/bash
# Code bytes, VGPRs, scratch and spills of every kernel of one HIP source (device code object of gfx950):
#   tools/kernel_code_size.sh semantic-embeddings_amd/csrc/rank_rows.hip [extra hipcc flags]
# The 64 KB instruction cache (shared by two CUs) is a hard wall for the per-row loop of the ranking kernels: profiles/r06_b_rank_icache.txt.
src=$1; shift
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o $tmp/k.co 2> $tmp/remarks.txt
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.elf
python3 - $tmp <<'PY'
import re, subprocess, sys
tmp = sys.argv[1]
res = {}
cur = None
for ln in open(tmp + "/remarks.txt"):
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1); res[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill): (\d+)", ln)
    if m and cur:
        res[cur][m.group(1).split()[0] + (" spill" if "Spill" in m.group(1) else "")] = int(m.group(2))
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-sW", tmp + "/k.elf"], capture_output=True, text=True).stdout
rows = []
for ln in out.splitlines():
    f = ln.split()
    if len(f) >= 8 and f[3] == "FUNC" and f[7] in res:
        if (int(f[2]), f[7]) not in rows:
            rows.append((int(f[2]), f[7]))
names = subprocess.run(["c++filt"], input="\n".join(n for _, n in rows), capture_output=True, text=True).stdout.splitlines()
for (size, mangled), name in sorted(zip(rows, names)):
    r = res[mangled]
    print("%7d B  vgpr %3d  scratch %4d  spills v%-3d s%-3d  %s" % (size, r.get("VGPRs", -1), r.get("ScratchSize", 0), r.get("VGPRs spill", 0), r.get("SGPRs spill", 0), name[:110]))
PY
rm -rf $tmp
