"""Build-time check of the hand-fetched quads in the rotation encoder (rotor.hip: prefetch_quads / quads_landed).

The next round's quads are loaded by hand into the wave's R highest registers (v(256-R)..v255, 8-wave kernels) and read out of
them, behind a hand-counted wait, by one later statement; the compiler knows the registers only as clobbered by both
statements.  The scheme is sound as long as the compiler itself never puts a value there, which it has no reason to (it
allocates upwards from v0 and these kernels need fewer than 240 registers) but is not forced to.  This script disassembles
the code objects inside the BUILT library (density_amd/libdensity_hip.so, or the path given: the shipped ISA, not a second compile)
and checks, for every 8-wave encoder instance, that
  * the only instructions naming a staging register are those loads (global_load_dword vN, ..) and the moves out of them
    (v_mov_b32 vX, vN), and
  * every run of moves directly follows an s_waitcnt vmcnt(..).
Also checked: the default decoder's stage B waits with vmcnt(12) and nothing in its round loop drains the memory queue; the
exchange stage kernels of exchange_stages.hip fit their 256 registers without scratch memory.
usage: python tools/check_isa.py [lib.so]   (exit code 1 on a violation; run by density_amd.build)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def regs_of(text):
    used = set(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        used |= set(range(int(a), int(b) + 1))
    return used

def check_function(name, rounds, body, top=256):
    stage = set(range(top - rounds, top))
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

LLVM = "/opt/rocm/lib/llvm/bin"

def shipped_code(lib):
    """Disassembly and kernel metadata of the code objects INSIDE the built library (what ships is what is checked: not a second
    compile with a compiler and flags of its own).  The .hip_fatbin section is a run of clang offload bundles, one per source file."""
    funcs, notes = {}, ""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
        for k, a in enumerate(starts):
            part, obj = os.path.join(tmp, f"b{k}.bin"), os.path.join(tmp, f"co{k}.o")
            open(part, "wb").write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + part, "--output=" + obj], check=True, stderr=subprocess.DEVNULL)
            if not os.path.exists(obj) or os.path.getsize(obj) == 0:
                continue
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", obj], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
                if m:
                    cur = funcs.setdefault(m.group(1), [])
                    continue
                t = line.split("//")[0].strip()
                if cur is not None and t and not t.startswith("."):
                    cur.append(t)
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], check=True, capture_output=True, text=True).stdout
    return funcs, notes

def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "density_amd", "libdensity_hip.so")
    if not os.path.exists(lib):
        print(f"check_isa: {lib} not built")
        return 1
    funcs, notes = shipped_code(lib)
    total = bad = 0
    for name, body in funcs.items():
        # every encoder instance with kept quads (last template argument): 8 waves stage in v(256-R)..v255, 12 waves in v(168-R)..v167
        ms = re.match(r"^_ZN7density20chameleon_encode_rotILi(\d+)ELi(\d+)ELb[01]ELb0ELb0ELb[01]ELb1EE", name)   # (<R, W, kProf, KEEP, EARLY, PAGED, SPLIT>: the split encoder)
        if ms:
            # sixteen waves share the CU: 128 registers each.  The common path of a chain wave must not touch scratch memory (a reload between the
            # token and the first exchange is hundreds of cycles of the critical section): from the ring's reads to the round's exchanges nothing but
            # what the rare paths (laid out in between) need — a handful of spills in all, none of them next to the exchanges
            R = int(ms.group(1))
            first = next((k for k, t in enumerate(body) if t.startswith("ds_mskor_rtn_b32")), None)
            near = [t for t in body[max(0, (first or 0) - 12):(first or 0) + R + 4] if t.startswith("scratch_")]
            spills = sum(t.startswith("scratch_") for t in body)
            if first is None or near or spills > 128:
                print(f"{name}: split encoder: scratch traffic at the exchanges {near} / {spills} scratch instructions in all")
                bad += 1
            # the emit waves' hand-issued loads (SGPR base + 32-bit offset): their destination registers are not named by any instruction before a
            # wait on the memory queue (the statement that claims them) — the compiler, which takes them for defined values, has moved or read none
            hand = 0
            for k, t in enumerate(body):
                m2 = re.match(r"^global_load_dword v(\d+), v\d+, s\[\d+:\d+\]", t)
                if not m2:
                    continue
                hand += 1
                x = int(m2.group(1))
                for u in body[k + 1:]:
                    if u.startswith("s_waitcnt") and "vmcnt(" in u:
                        break
                    if x in regs_of(u) and not re.match(r"^global_load_dword v\d+, v\d+, s\[", u):
                        print(f"{name}: v{x}, in flight since `{t}`, is named by `{u}` before any wait on the memory queue")
                        bad += 1
                        break
            if not hand:
                print(f"{name}: split encoder: no hand-issued loads found")
                bad += 1
            total += hand
            continue
        m = re.match(r"^_ZN7density20chameleon_encode_rotILi(\d+)ELi(\d+)ELb[01]ELb1ELb[01]ELb[01]ELb0EE", name)   # (<R, W, kProf, KEEP, EARLY, PAGED, SPLIT>)
        if not m:
            continue
        g, b = check_function(name, int(m.group(1)), body, 256 if int(m.group(2)) == 8 else 168)
        total += g; bad += b
    # the decoder's exact waits: stage B must wait for "all but the last 12 (stores)", not for everything (a branch around the record stores,
    # a load left pending across the loop head ... turn it into vmcnt(0) and cost 5-10 % without a test failing)
    dec = 0
    for name, body in funcs.items():
        m = re.match(r"^_ZN7density20chameleon_decode_rotILi(\d+)ELi(\d+)ELb0E", name)       # every shipped geometry (rounds of R records: R stores per round)
        if not m or int(m.group(1)) < 12:
            continue
        want = f"s_waitcnt vmcnt({m.group(1)})"
        first = next(k for k, t in enumerate(body) if t.startswith("ds_mskor_rtn_b32"))
        window = body[max(0, first - 75 * int(m.group(1))):first]
        waits = [t for t in window if t.startswith("s_waitcnt vmcnt(") and "lgkmcnt" not in t]
        if want not in waits:
            print(f"{name}: stage B no longer waits with {want}: {waits}")
            bad += 1
        elif "s_waitcnt vmcnt(0)" in waits[waits.index(want):]:
            print(f"{name}: a full drain (vmcnt(0)) inside the round loop: {waits}")
            bad += 1
        # ... and no scratch memory anywhere in it (a spilled loop invariant is reloaded — and waited for with vmcnt(0) — inside the critical section)
        # (one store at the kernel's start — the epilogue's in-order state — is not a spill; any reload is)
        n_ld, n_st = sum(t.startswith("scratch_load") for t in body), sum(t.startswith("scratch_store") for t in body)
        if n_ld > 1 or n_st > 1:
            print(f"{name}: spills to scratch memory ({n_st} stores, {n_ld} loads)")
            bad += 1
        dec += 1
    if not dec:
        print("check_isa: decoder instance not found")
        bad += 1
    # exchange_stages.hip: eight waves of a stage work-group share a CU, two per SIMD — 256 registers each.  A stage kernel that needs
    # more spills to scratch memory inside the token's critical path without a test failing.
    stages = 0
    for m in re.finditer(r"\.name:\s+(\S*exchange_stage\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", notes):
        stages += 1
        if int(m.group(2)) or int(m.group(4)) or int(m.group(3)) > 256:
            print(f"{m.group(1)}: scratch {m.group(2)} bytes, {m.group(3)} registers, {m.group(4)} spilled")
            bad += 1
    if not stages:
        print("check_isa: no exchange stage kernels found")
        bad += 1
    print(f"check_isa: {os.path.basename(lib)}: {total} hand-issued loads in the encoder instances with kept quads, decoder waits checked, {stages} exchange stage kernels without scratch, {bad} violation(s)")
    return 1 if bad or not total else 0

if __name__ == "__main__":
    sys.exit(main())
