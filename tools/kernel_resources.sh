#!/bin/bash
# VGPR / SGPR / scratch / LDS of every kernel of a HIP source (compiles the device side only)
#   tools/kernel_resources.sh [source=rmcl_amd/csrc/kernels.hip] [name filter]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${1:-$ROOT/rmcl_amd/csrc/kernels.hip}
OUT=/tmp/kres_$$
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function --cuda-device-only \
  -I$ROOT/rmcl_amd/csrc -c "$SRC" -o $OUT/dev.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$OUT/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$OUT/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT/dev.co > $OUT/notes.txt
python3 - "$OUT/notes.txt" "${2:-}" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2]
for blk in t.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt and flt not in name:
        continue
    g = lambda k: (re.findall(k + r":\s+(\d+)", blk) or ["?"])[0]
    print("%-110s vgpr %3s sgpr %3s scratch %4s lds %6s" % (name[-110:], g(r"\.vgpr_count"), g(r"\.sgpr_count"),
          g(r"\.private_segment_fixed_size"), g(r"\.group_segment_fixed_size")))
PY
cp $OUT/dev.co /tmp/last_dev.co
rm -rf $OUT
