"""Audit of compiler-scheduled fp64 MFMAs in the built code object (gfx950).

v_mfma_f64_16x16x4_f64 is a 16-pass instruction on gfx950.  Measured (tools/mfma_hazard_probe.hip,
profiles/r01_probe_mfma_hazard.txt): a VALU read of the first three result pairs is interlocked by the hardware,
a read of the LAST pair (destination registers 6-7) sees stale data unless it is at least 18 wait states behind
the MFMA or comes after an (interlocked, hence stalling) read of one of the other pairs -- and hipcc (ROCm 7.2)
only leaves the 11 (or fewer) wait states that the 8-pass gfx942 instruction needs.  This
walks the disassembly and reports every MFMA whose last result pair is touched within WINDOW wait states on the
path (branches followed; every instruction counted as one wait state, s_nop N as N + 1).
Accumulating MFMAs (same registers as C operand and destination) are exempt: the pipe interlocks those.

usage: python tools/mfma_hazard_audit.py [path/to/scaml_gfx950.hsaco]
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd", "lib", "build_libscaml_hip", "scaml_gfx950.hsaco")
WINDOW = 18
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", path], capture_output=True, text=True).stdout.split("\n")
reg = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
def regs(s):
    out = set()
    for m in reg.finditer(s):
        if m.group(3) is not None: out.add(int(m.group(3)))
        else: out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
func, bad, total = "?", [], 0
body, at = [], {}
for ln in txt:
    m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
    if m: func = m.group(1); continue
    if "//" in ln:
        ins, cm = ln.split("//", 1)
        addr = int(cm.split(":")[0].strip(), 16)
        at[addr] = len(body)
        body.append((func, ins.strip(), addr))

def walk(i, ws, dst, full, fn, seen, hits):
    """follow every path from instruction i until WINDOW wait states have passed"""
    while ws < WINDOW and i < len(body) and (i, ws) not in seen:
        seen.add((i, ws))
        fn2, nxt, addr = body[i]
        if fn2 != fn: return
        name = nxt.split()[0]
        if name in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"): return
        if name.startswith("s_cbranch") or name == "s_branch":
            off = int(nxt.split()[1]); off -= 65536 if off >= 32768 else 0
            tgt = at.get(addr + 4 + 4 * off)
            if tgt is not None: walk(tgt, ws + 1, dst, full, fn, seen, hits)
            if name == "s_branch": return
            ws += 1; i += 1; continue
        if name == "s_nop":
            ws += int(nxt.split()[1]) + 1; i += 1; continue
        if name.startswith("v_mfma"):
            o = nxt.split(None, 1)[1].split(",")
            if (regs(o[1]) | regs(o[2])) & dst: hits.append((ws, nxt))
            if regs(o[0]) == full: return   # accumulates into the same tile: a new window starts at that MFMA
        elif name[:2] in ("v_", "ds") or name.startswith(("global_", "scratch_", "buffer_", "flat_")):
            ops = nxt.split(None, 1)[1] if " " in nxt else ""
            if "_load_" in name or name.startswith("ds_read"):
                # the destination of a memory load is written when the data returns, hundreds of cycles after every pass of the MFMA
                # has retired: a pure write-after-write, not a read of the pair.  Only the address / data operands count.
                ops = ops.split(",", 1)[1] if "," in ops else ""
            touched = regs(ops)
            if touched & dst:
                hits.append((ws, nxt)); return
            if name[:2] == "v_" and touched & (full - dst): return   # interlocked read: stalls until the MFMA is done
        ws += 1; i += 1

for i, (fn, ins, addr) in enumerate(body):
    if not ins.startswith("v_mfma_f64_16x16x4"): continue
    full = regs(ins.split(None, 1)[1].split(",")[0])
    if not full: continue   # AGPR destination: hand-managed tiles, waits are explicit in the source
    total += 1
    dst = {max(full) - 1, max(full)}   # only the last pair is not interlocked
    hits = []
    walk(i + 1, 0, dst, full, fn, set(), hits)
    for ws, nxt in hits: bad.append((fn, i, ins, ws, nxt))
print(f"{total} fp64 MFMAs with a VGPR destination; {len(bad)} reads of a last result pair within {WINDOW} wait states (all paths followed)")
for fn, i, ins, ws, nxt in bad: print(f"  HAZARD {fn[:60]} #{i}: {ins}  --{ws} ws-->  {nxt}")
sys.exit(1 if bad else 0)
