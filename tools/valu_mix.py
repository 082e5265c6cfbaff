"""Static fp32-operation count of a kernel's hot loop from its gfx950 assembly (the figure behind bench.py's
`roofline.valu.flops_per_valu_inst`):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Ipings_amd/csrc \
        -DPINGS_BUILDING_DLL --save-temps -c pings_amd/csrc/raster_bwd.hip -o /tmp/rb.o
  python tools/valu_mix.py raster_bwd-hip-amdgcn-amd-amdhsa-gfx950.s blend_bwd_kernelILi0ELi4E
The loop is the innermost one that holds the kernel's v_exp_f32; an fma counts two operations, a packed instruction two lanes."""
import re
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
a = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
body = lines[a:b]
ex = [i for i, l in enumerate(body) if "v_exp_f32" in l]
start = ex[0]
while not re.match(r"^\.LBB\d+_\d+:", body[start]):
    start -= 1
label = body[start].split(":")[0]
end = ex[-1]
while not ("s_cbranch" in body[end] and label in body[end]) and end < len(body) - 1:
    end += 1
ops = [l.split()[0] for l in body[start:end + 1] if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;") and l.split()]
valu = [o for o in ops if o.startswith("v_")]
flops = 0
for o in valu:
    w = 2 if o.startswith("v_pk_") else 1
    if re.match(r"v_(pk_)?(fma|fmac|mac|mad)_f32", o):
        flops += 2 * w
    elif re.match(r"v_(pk_)?(add|sub|subrev|mul|max|min)_f32", o) or re.match(r"v_(exp|rcp|log|rsq|sqrt)_f32", o):
        flops += w
print(f"loop {label}: {len(ops)} instructions, {len(valu)} vector, {flops} fp32 operations per lane and trip = "
      f"{flops / len(valu):.3f} per vector instruction")
print(Counter(valu).most_common(12))
