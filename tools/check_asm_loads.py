"""Static check of hand-waited asm loads in a kernel's ISA (hipcc -S output): no instruction may touch the destination (or reuse it as an address) of an
asm `global_load_dwordx4` before an `s_waitcnt vmcnt(N)` that retires it.  Usage: hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S -o f32.s
lemas_tts_amd/csrc/gemm_f32.hip -I include -I lemas_tts_amd/csrc && python tools/check_asm_loads.py f32.s   (csrc/gemm_f32.hip: two register sets in flight)"""
import re, sys
f = sys.argv[1]
pending = {}   # reg -> line of load
kern = None
bad = 0
inasm = False
nload = 0
for ln, line in enumerate(open(f), 1):
    t = line.strip()
    if t.endswith(":") and t.startswith("_Z"):
        kern = t[:-1]; pending = {}
    if t.startswith(";;#ASMSTART"): inasm = True; continue
    if t.startswith(";;#ASMEND"): inasm = False; continue
    if not t or t.startswith(";") or t.startswith("."): continue
    if t.startswith("s_endpgm"): pending = {}; continue
    regs = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", t):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", t):
        regs.add(int(m.group(1)))
    if inasm and t.startswith("global_load_dwordx4"):
        m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]", t)
        dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
        src = set(range(int(m.group(3)), int(m.group(4)) + 1))
        hit = src & set(pending)
        if hit: print("BAD(addr)", kern, ln, t); bad += 1
        for r in dst: pending[r] = ln
        nload += 1
        continue
    if t.startswith("s_waitcnt") and "vmcnt" in t:
        # exact: vmcnt(N) retires all but the youngest N loads
        m = re.search(r"vmcnt\((\d+)\)", t)
        n = int(m.group(1))
        lines = sorted(set(pending.values()))
        keep = set(lines[len(lines) - n:]) if n > 0 else set()
        pending = {r: l for r, l in pending.items() if l in keep}
        continue
    hit = regs & set(pending)
    if hit:
        print("BAD", kern[:60] if kern else None, ln, t, sorted(hit)[:4]); bad += 1
print("asm loads:", nload, "violations:", bad)
