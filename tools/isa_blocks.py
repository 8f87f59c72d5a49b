"""dev: basic-block table of one kernel in a hipcc -S dump: per block the VALU / SALU / LDS / memory instruction counts, scratch traffic,
v_readlane / v_writelane (SGPR spills parked in VGPR lanes) and the branch targets -- to see what a model step of the rollout kernel is made of.
usage: python tools/isa_blocks.py file.s <mangled-kernel-name-substring> [--dump LABEL]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith(pat) or (l.startswith("_Z") and pat in l.split(":")[0] and ":" in l))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
blocks, cur = [], {"label": "entry", "ins": []}
for l in src[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur); cur = {"label": m.group(1), "ins": []}
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append(cur); cur = {"label": m.group(1), "ins": []}
        continue
    cur["ins"].append(t.split(";")[0].strip())
blocks.append(cur)
if "--dump" in sys.argv:
    lab = sys.argv[sys.argv.index("--dump") + 1]
    for b in blocks:
        if b["label"] == lab:
            print("\n".join(b["ins"]))
    sys.exit(0)
tot = dict(valu=0, salu=0)
print("%-12s %5s %5s %5s %4s %4s %4s %4s %4s  %s" % ("block", "VALU", "f64", "SALU", "LDS", "glb", "scr", "rdl", "wrl", "branches"))
for b in blocks:
    ins = b["ins"]
    op = [i.split()[0] for i in ins]
    valu = [o for o in op if o.startswith("v_") and not o.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))]
    f64 = [o for o in valu if "f64" in o]
    salu = [o for o in op if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_sleep", "s_setprio"))]
    lds = [o for o in op if o.startswith("ds_")]
    glb = [o for o in op if o.startswith(("global_", "flat_", "buffer_", "s_load", "s_buffer"))]
    scr = [o for o in op if o.startswith("scratch_")]
    rdl = [o for o in op if o.startswith("v_readlane")]
    wrl = [o for o in op if o.startswith("v_writelane")]
    br = [i.split()[-1] for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    print("%-12s %5d %5d %5d %4d %4d %4d %4d %4d  %s" % (b["label"], len(valu), len(f64), len(salu), len(lds), len(glb), len(scr), len(rdl), len(wrl), " ".join(br)))
