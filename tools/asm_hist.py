"""Instruction-mix histogram along the ISA stream of one kernel (from hipcc -save-temps .s)."""
import re, sys
path, pick = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "Li16E"
lines = open(path).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_ZN5scaml.*:\s*(;.*)?$', l) and pick in l]
s0 = starts[-1]
e0 = next(i for i in range(s0, len(lines)) if '.amdhsa_kernel' in lines[i])
body = lines[s0:e0]
open('/tmp/kernel_pick.s', 'w').write('\n'.join(body))
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 400
pats = [('scr_st', 'scratch_store'), ('scr_ld', 'scratch_load'), ('mfma', 'v_mfma'), ('ldexp', 'v_ldexp'), ('g_st', 'global_store'), ('g_ld', 'global_load'),
        ('bar', 's_barrier'), ('ds_r', 'ds_read'), ('ds_w', 'ds_write'), ('rl', 'v_readlane'), ('wl', 'v_writelane'), ('mov', r'v_mov_b'), ('fma', 'v_fma_f64'), ('wait', 's_waitcnt')]
for i in range(0, len(body), chunk):
    seg = '\n'.join(body[i:i + chunk])
    print(f"{i:6d} " + ' '.join(f"{n}={len(re.findall(p, seg)):3d}" for n, p in pats))
print(len(body), lines[s0][:80])
