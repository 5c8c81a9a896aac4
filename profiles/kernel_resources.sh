#!/bin/bash
# Register / scratch use of every rt_trace instantiation, read from the code object's metadata notes.
#   bash profiles/kernel_resources.sh [object file]     (default: the product build, rt_kernel_fast.o)
set -e
B=/opt/rocm/lib/llvm/bin
OBJ=${1:-$(dirname "$0")/../html5-canvas-raytracer_amd/csrc/rt_kernel_fast.o}
T=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$T/k.bin "$OBJ"
$B/clang-offload-bundler --unbundle --type=o --input=$T/k.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
$B/llvm-readelf --notes $T/k.co | python3 -c '
import re, sys
print("rt_trace<REFRACT,COUNT,SS2,GRID[,W1]>  vgpr  vgpr_spill  sgpr  sgpr_spill  scratch_bytes")
for block in re.split(r"\n\s+- \.agpr_count:", sys.stdin.read())[1:]:
    f = dict(re.findall(r"\.(name|vgpr_count|vgpr_spill_count|sgpr_count|sgpr_spill_count|private_segment_fixed_size):\s+(\S+)", block))
    m = re.search(r"rt_traceILb([01])ELb([01])ELb([01])ELb([01])ELb([01])E", f.get("name", ""))
    if m:
        print("  <%s>                     %4s  %10s  %4s  %10s  %13s" % (",".join(m.groups()), f["vgpr_count"], f["vgpr_spill_count"], f["sgpr_count"], f["sgpr_spill_count"], f["private_segment_fixed_size"]))
'
rm -rf $T
