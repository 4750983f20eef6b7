"""Give every function defined in a device LLVM-IR file its own attribute group and set the register-file
attributes clang has no source spelling for.

  "amdgpu-agpr-alloc"="0"   the compiler allocates NO AGPRs: the accumulator half of the unified
                            register file belongs to the hand-written asm in the kernels
                            (csrc/gp_fit_fused.hip); with any other value hipcc uses free AGPRs
                            as VGPR spill space between asm statements and corrupts the tiles
  "amdgpu-num-vgpr"="V/2"   caps the arch VGPRs at V (LLVM doubles this attribute when it sizes
                            the unified file), so that VGPRs + hand-managed AGPRs <= 256 where a
                            workgroup needs 2 waves per SIMD

usage: patch_ir.py in.ll out.ll  [substring=VGPR_CAP ...]
       e.g. patch_ir.py dev.ll dev.patched.ll Li16ELi7E=96
"""
import re
import sys


def patch(text: str, caps: dict) -> str:
    groups = dict(re.findall(r"^attributes (#\d+) = \{ (.*) \}$", text, flags=re.M))
    next_id = max(int(g[1:]) for g in groups) + 1
    out_defs = []
    new_groups = []

    def repl(m):
        nonlocal next_id
        line, name, grp = m.group(0), m.group(1), m.group(2)
        attrs = groups[grp]
        attrs = re.sub(r' ?"amdgpu-agpr-alloc"="[^"]*"', "", attrs)
        attrs = re.sub(r' ?"amdgpu-num-vgpr"="[^"]*"', "", attrs)
        attrs += ' "amdgpu-agpr-alloc"="0"'
        for sub, cap in caps.items():
            if sub in name:
                attrs += f' "amdgpu-num-vgpr"="{cap // 2}"'
        gid = f"#{next_id}"
        next_id += 1
        new_groups.append(f"attributes {gid} = {{ {attrs} }}")
        out_defs.append(name)
        return line[: m.start(2) - m.start(0)] + gid + line[m.end(2) - m.start(0):]

    # every function DEFINITION (kernels and the non-inlined device functions they call)
    text = re.sub(r"^define [^\n]*@([\w.$]+)\([^\n]*? (#\d+)(?= )[^\n]*\{$", repl, text, flags=re.M)
    if not out_defs:
        raise SystemExit("patch_ir: no function definitions found")
    return text.rstrip("\n") + "\n" + "\n".join(new_groups) + "\n", out_defs


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    caps = {}
    for a in sys.argv[3:]:
        k, v = a.split("=")
        caps[k] = int(v)
    patched, names = patch(open(src).read(), caps)
    open(dst, "w").write(patched)
    print(f"patch_ir: {len(names)} functions patched")
