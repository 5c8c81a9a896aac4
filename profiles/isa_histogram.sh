#!/bin/bash
# Static ISA histogram (instructions by mnemonic) of one rt_trace instantiation of the product build, from the code object.
#   bash profiles/isa_histogram.sh [REFRACT COUNT SS2 GRID]      default 0 0 0 0 = the headline kernel
set -e
B=/opt/rocm/lib/llvm/bin
R=${1:-0}; C=${2:-0}; S=${3:-0}; G=${4:-0}
OBJ=$(dirname "$0")/../html5-canvas-raytracer_amd/csrc/rt_kernel_fast.o
T=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$T/k.bin "$OBJ"
$B/clang-offload-bundler --unbundle --type=o --input=$T/k.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
$B/llvm-objdump -d $T/k.co > $T/k.s
SYM="rt_traceILb${R}ELb${C}ELb${S}ELb${G}E"
awk -v sym="$SYM" '$0 ~ sym && /^[0-9a-f]+ </ {f=1; next} f && /^[0-9a-f]+ </ {exit} f {print}' $T/k.s | awk 'NF && $1 !~ /:$/ {print $1}' > $T/m.txt
echo "rt_trace<$R,$C,$S,$G> (product build, gfx950): $(wc -l < $T/m.txt) instructions (static; dynamic counts: the SQ_INSTS_* counters in the rocprof summaries)"
python3 - "$T/m.txt" <<'PY'
import sys, collections
m = [l.strip() for l in open(sys.argv[1]) if l.strip()]
c = collections.Counter(m)
def cls(x):
    if x.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_rsq_f64", "v_rcp_f64", "v_rndne_f64", "v_ldexp_f64", "v_div_", "v_trunc_f64", "v_floor_f64", "v_ceil_f64", "v_fract_f64", "v_frexp", "v_sqrt_f64")): return "VALU fp64 arithmetic"
    if x.startswith("v_cmp") and "f64" in x: return "VALU fp64 compare"
    if x.startswith("v_cvt"): return "VALU convert"
    if x.startswith(("v_mov", "v_cndmask", "v_readlane", "v_writelane", "v_readfirstlane", "v_accvgpr")): return "VALU move/select"
    if x.startswith("v_"): return "VALU integer/other"
    if x.startswith(("s_load", "s_buffer_load")): return "SMEM"
    if x.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_waitcnt", "s_nop", "s_setpc", "s_swappc", "s_getpc", "s_sleep")): return "control/wait"
    if x.startswith("s_"): return "SALU"
    if x.startswith("ds_"): return "LDS"
    if x.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    return "other"
k = collections.Counter()
for x, n in c.items(): k[cls(x)] += n
print("\nby class:")
for x, n in k.most_common(): print("  %-24s %5d  %5.1f %%" % (x, n, 100.0 * n / len(m)))
print("\nby mnemonic:")
for x, n in c.most_common(): print("  %-28s %5d" % (x, n))
PY
rm -rf $T
