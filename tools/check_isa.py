"""Build-time check of the hand-fetched quads in the rotation encoder (rotor.hip: prefetch_quads / quads_landed).

The next round's quads are loaded by hand into the wave's R highest registers (v(256-R)..v255, 8-wave kernels) and read out of
them, behind a hand-counted wait, by one later statement; the compiler knows the registers only as clobbered by both
statements.  The scheme is sound as long as the compiler itself never puts a value there, which it has no reason to (it
allocates upwards from v0 and these kernels need fewer than 240 registers) but is not forced to.  This script compiles
rotor.hip to assembly and checks, for every 8-wave encoder instance, that
  * the only instructions naming a staging register are those loads (global_load_dword vN, ..) and the moves out of them
    (v_mov_b32 vX, vN), and
  * every run of moves directly follows an s_waitcnt vmcnt(..).
usage: python tools/check_isa.py   (exit code 1 on a violation; run by density_amd.build)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def regs_of(text):
    used = set(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        used |= set(range(int(a), int(b) + 1))
    return used

def check_function(name, rounds, body):
    stage = set(range(256 - rounds, 256))
    is_move = lambda t: bool(re.match(r"^v_mov_b32(?:_e32)? v(\d+), v(\d+)$", t)) and int(t.split("v")[-1]) in stage and int(re.match(r"^\S+ v(\d+)", t).group(1)) not in stage
    loads = moves = bad = 0
    for k, t in enumerate(body):
        if not regs_of(t) & stage:
            continue
        m = re.match(r"^global_load_dword v(\d+), v\[\d+:\d+\], off", t)
        if m and int(m.group(1)) in stage:
            loads += 1
            continue
        if is_move(t):
            moves += 1
            p = k - 1
            while is_move(body[p]):
                p -= 1
            if not body[p].startswith("s_waitcnt vmcnt("):
                print(f"{name}: moves out of the staging registers without a wait in front: {body[p]} / {t}")
                bad += 1
            continue
        print(f"{name}: a staging register is named by: {t}")
        bad += 1
    if not loads or not moves:
        print(f"{name}: no hand-issued loads found")
        bad += 1
    return loads, bad

def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "rotor.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "density_amd", "csrc", "rotor.hip")], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    total = bad = 0
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN7density20chameleon_encode_rotILi(\d+)ELi8ELb[01]E\w*):", lines[i])
        if not m:
            i += 1
            continue
        j = i
        while not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = [l.split(";")[0].strip() for l in lines[i:j]]
        body = [t for t in body if t and not t.startswith(".")]
        g, b = check_function(m.group(1), int(m.group(2)), body)
        total += g; bad += b
        i = j
    print(f"check_isa: {total} hand-issued loads in the 8-wave encoder instances, {bad} violation(s)")
    return 1 if bad or not total else 0

if __name__ == "__main__":
    sys.exit(main())
