"""Build-time check of the hand-fetched quads in the rotation encoder (rotor.hip: prefetch_quads / quads_landed).

The next round's quads are loaded by hand into the wave's R highest registers (v(256-R)..v255, 8-wave kernels) and read out of
them, behind a hand-counted wait, by one later statement; the compiler knows the registers only as clobbered by both
statements.  The scheme is sound as long as the compiler itself never puts a value there, which it has no reason to (it
allocates upwards from v0 and these kernels need fewer than 240 registers) but is not forced to.  This script compiles
rotor.hip to assembly and checks, for every 8-wave encoder instance, that
  * the only instructions naming a staging register are those loads (global_load_dword vN, ..) and the moves out of them
    (v_mov_b32 vX, vN), and
  * every run of moves directly follows an s_waitcnt vmcnt(..).
Also checked: the default decoder's stage B waits with vmcnt(12) and nothing in its round loop drains the memory queue; the
exchange stage kernels of exchange_stages.hip fit their 256 registers without scratch memory.
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
    # the decoder's exact waits: stage B must wait for "all but the last 12 (stores)", not for everything (a branch around the record stores,
    # a load left pending across the loop head ... turn it into vmcnt(0) and cost 5-10 % without a test failing)
    i = 0
    dec = 0
    while i < len(lines):
        m = re.match(r"^(_ZN7density20chameleon_decode_rotILi12ELi12ELb0E\w*):", lines[i])
        if not m:
            i += 1
            continue
        j = i
        while not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = [l.split(";")[0].strip() for l in lines[i:j]]
        body = [t for t in body if t and not t.startswith(".")]
        first = next(k for k, t in enumerate(body) if t.startswith("ds_mskor_rtn_b32"))
        window = body[max(0, first - 900):first]
        waits = [t for t in window if t.startswith("s_waitcnt vmcnt(") and "lgkmcnt" not in t]
        if "s_waitcnt vmcnt(12)" not in waits:
            print(f"{m.group(1)}: stage B no longer waits with vmcnt(12): {waits}")
            bad += 1
        elif "s_waitcnt vmcnt(0)" in waits[waits.index("s_waitcnt vmcnt(12)"):]:
            print(f"{m.group(1)}: a full drain (vmcnt(0)) inside the round loop: {waits}")
            bad += 1
        dec += 1
        i = j
    if not dec:
        print("check_isa: decoder instance not found")
        bad += 1
    # exchange_stages.hip: eight waves of a stage work-group share a CU, two per SIMD — 256 registers each.  A stage kernel that needs
    # more spills to scratch memory inside the token's critical path without a test failing.
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "stages.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "density_amd", "csrc", "exchange_stages.hip")], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    stages = 0
    for m in re.finditer(r"\.name:\s+(\S*exchange_stage\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
        stages += 1
        if int(m.group(2)) or int(m.group(4)) or int(m.group(3)) > 256:
            print(f"{m.group(1)}: scratch {m.group(2)} bytes, {m.group(3)} registers, {m.group(4)} spilled")
            bad += 1
    if not stages:
        print("check_isa: no exchange stage kernels found")
        bad += 1
    print(f"check_isa: {total} hand-issued loads in the 8-wave encoder instances, decoder waits checked, {stages} exchange stage kernels without scratch, {bad} violation(s)")
    return 1 if bad or not total else 0

if __name__ == "__main__":
    sys.exit(main())
