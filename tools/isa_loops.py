#!/usr/bin/env python3
"""Static instruction counts of the step-loop bodies of a kernel (segments between s_barrier).

    python tools/isa_loops.py neural-speech-decoding_amd/csrc/nsd_lstm2_bwd48.hip [kernel-name-substring]

Measured on MI355X: a wave issues roughly one instruction per ~5 cycles whatever its ILP, so the instruction
count of the slowest wave's loop body is the first-order model of a step's duration."""
import collections, re, subprocess, sys, tempfile, os

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
out = tempfile.mktemp(suffix=".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                       "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
os.unlink(out)
# kernels: from label line "<name>:" to s_endpgm
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if not m or pat not in m.group(1):
        i += 1
        continue
    name = m.group(1)
    j = i
    while j < len(lines) and not lines[j].startswith(".Lfunc_end"):      # a role-split kernel has one s_endpgm per role
        j += 1
    body = [l.strip() for l in lines[i:j] if l.strip() and not l.strip().startswith((";", "."))]
    print(f"== {name}: {len(body)} instructions")
    seg, segs = [], []
    for l in body:
        seg.append(l)
        if l.startswith("s_barrier"):
            segs.append(seg); seg = []
    for k, sg in enumerate(segs):
        ops = [l.split()[0] for l in sg]
        c = collections.Counter()
        for o in ops:
            if o.startswith("v_mfma"): c["mfma"] += 1
            elif o.startswith("v_pk_fma"): c["pk_fma"] += 1
            elif o.startswith("v_"): c["valu"] += 1
            elif o.startswith("ds_"): c["lds"] += 1
            elif o.startswith(("global_", "buffer_", "scratch_")): c["vmem"] += 1
            elif o.startswith(("s_waitcnt", "s_nop")): c["wait/nop"] += 1
            elif o.startswith("s_"): c["salu"] += 1
        tags = [t for t, p in (("exp", "v_exp"), ("rcp", "v_rcp"), ("dpp", "dpp")) if any(p in l for l in sg)]
        if len(sg) > 25:
            print(f"  segment {k:2d}: {len(sg):4d} instr  " + "  ".join(f"{a}={b}" for a, b in sorted(c.items())) + "  " + ",".join(tags))
    i = j
