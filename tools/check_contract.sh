#!/bin/bash
# dev: the model code of car_dynamics.h must leave the compiler nothing to contract (see "Rounding discipline" there): compile the rollout
# kernels under -ffp-contract=fast (the product build) and =off and compare the instruction streams of the 1-car kernel.  Expected: differences
# only inside inlined libm code of the cold paths (sin / cos / fmod: v_trig_preop, v_ldexp, v_fract neighbourhoods), none in the sub-step loop.
cd "$(dirname "$0")/.."
for f in fast off; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=$f -S --cuda-device-only mpopis_amd/csrc/kernels_rollout.hip -Iinclude -Impopis_amd/csrc -o /tmp/kr_$f.s 2>/dev/null || exit 1
done
python3 - <<'PY'
import re, difflib
def func(path, name):
    out, cur = [], False
    for l in open(path):
        if l.startswith(name + ':'): cur = True; continue
        if cur and l.startswith('.Lfunc_end'): break
        if cur and (l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;') or l.startswith('.LBB')): out.append(l.rstrip())
    return out
n = '_ZN6mpopis13k_rollout_carILi1ELi4ELb0ELb1EEEvNS_11RolloutArgsE'
a, b = func('/tmp/kr_fast.s', n), func('/tmp/kr_off.s', n)
op = lambda L: [x.split()[0] for x in L]
libm = [i for i, l in enumerate(a) if any(t in l for t in ('v_trig_preop', 'v_fract_f64', '0x54442d18', '0x3ff921fb', '0xbff921fb'))]
near_libm = lambda i: any(abs(i - j) <= 150 for j in libm)
print("instructions: fast %d, off %d" % (len(a), len(b)))
nd = own = 0
for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, op(a), op(b), autojunk=False).get_opcodes():
    if tag != 'equal':
        nd += 1
        if not near_libm(i1):
            own += 1
            print("MODEL CODE: %s fast[%d:%d] off[%d:%d] %s | %s" % (tag, i1, i2, j1, j2, [x.strip() for x in a[i1:i2][:3]], [x.strip() for x in b[j1:j2][:3]]))
print("%d differing regions, %d outside inlined libm code (must be 0)" % (nd, own))
raise SystemExit(1 if own else 0)
PY
