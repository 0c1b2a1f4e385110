#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel's hot loop, priced with measured issue costs (dev aid; runs without a GPU).

    python tools/isa_mix.py [--costs profiles/<tag>_valu_microbench.json] [--json out.json] [kernel-name-substring ...]

Compiles the translation units of gym_pomdp_amd/csrc/ for gfx950 to assembly text (hipcc --cuda-device-only -S), finds each
requested kernel, takes its loop with the most instructions of its own (the step loop of the fused kernels, the four-step
loop of the rollout / heuristic kernels) without the loops nested inside it (they are the 2^-27 tie paths and the
continuation passes), and counts its VALU instructions by mnemonic.  With a cost table (tools/valu_microbench: shader cycles
per wave64 instruction per SIMD) the mix gives the average issue cost of the loop's vector instructions — what turns a
measured wave-instructions/s figure into a fraction of the SIMDs' issue cycles:

    issue peak [wave-instructions/s] = 256 CUs x 4 SIMDs x 2.4e9 Hz / (average cycles per instruction of this mix)

A static mix weighs every instruction of the loop body once; exec-masked side paths (rare) weigh like the common path.
"""
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "gym_pomdp_amd", "csrc")
UNITS = ["fused_rock.hip", "fused_tag.hip", "fused_battleship.hip", "fused_misc.hip", "planner.hip"]   # where the kernels of interest live
DEFAULT = ["steps_quad_kernel<pomdp::RockEnv<1, false>, ", "steps_quad_kernel<pomdp::RockEnv<2, false>, ", "rollout_kernel<pomdp::RockEnv<2, false>",
           "rollout_kernel<pomdp::RockEnv<1, false>", "rollout_kernel<pomdp::TagEnv", "heuristic_steps_kernel<pomdp::RockEnv<1, false>, false",
           "heuristic_steps_kernel<pomdp::RockEnv<2, false>, false", "heuristic_steps_kernel<pomdp::TagEnv, false",
           "tag_steps_quad_kernel<true, ", "network_steps_quad_kernel<2, ", "steps_quad_generic_kernel<pomdp::TigerEnv, ",
           "steps_kernel<pomdp::RockEnv<1, false>, 1, true, true, ", "steps_kernel<pomdp::RockEnv<1, false>, 2, true, false, ", "battleship_steps_quad_kernel<4, ", "battleship_steps_quad_kernel<1, "]

# mnemonic -> the measured class that prices it (tools/valu_microbench.hip op names); anything else: DEFAULT_COST
ALIAS = {
    "v_add_u32": "v_add_u32", "v_sub_u32": "v_sub_u32", "v_subrev_u32": "v_sub_u32", "v_add_i32": "v_add_u32",
    "v_and_b32": "v_and_b32", "v_or_b32": "v_or_b32", "v_xor_b32": "v_xor_b32", "v_not_b32": "v_mov_b32", "v_bfi_b32": "v_and_or_b32",
    "v_bitop3_b32": "v_bitop3_b32", "v_cndmask_b32": "v_cndmask_b32",
    "v_lshrrev_b32": "v_lshrrev_b32", "v_lshlrev_b32": "v_lshrrev_b32", "v_ashrrev_i32": "v_lshrrev_b32",
    "v_lshl_or_b32": "v_lshl_or_b32", "v_and_or_b32": "v_and_or_b32", "v_or3_b32": "v_or3_b32", "v_add3_u32": "v_add3_u32",
    "v_lshl_add_u32": "v_lshl_add_u32", "v_add_lshl_u32": "v_lshl_add_u32", "v_xad_u32": "v_xad_u32",
    "v_mul_hi_u32": "v_mul_hi_u32", "v_mul_lo_u32": "v_mul_lo_u32", "v_mad_u64_u32": "v_mad_u64_u32", "v_mul_u32_u24": "v_mad_u32_u24",
    "v_mad_u32_u24": "v_mad_u32_u24", "v_mul_hi_u32_u24": "v_mad_u32_u24",
    "v_sad_u8": "v_sad_u8", "v_bfe_u32": "v_bfe_u32", "v_bfe_i32": "v_bfe_u32", "v_mov_b32": "v_mov_b32", "v_bcnt_u32_b32": "v_bcnt_u32_b32",
    "v_alignbit_b32": "v_alignbit_b32", "v_min_u32": "v_min_u32", "v_max_u32": "v_min_u32", "v_min_i32": "v_min_u32", "v_max_i32": "v_min_u32",
    "v_min3_u32": "v_add3_u32", "v_perm_b32": "v_perm_b32", "v_lshlrev_b64": "v_lshlrev_b64", "v_lshrrev_b64": "v_lshlrev_b64",
    "v_add_co_u32": "v_add_co_u32", "v_addc_co_u32": "v_add_co_u32", "v_sub_co_u32": "v_add_co_u32", "v_subb_co_u32": "v_add_co_u32",
    "v_add_f64": "v_add_f64", "v_mul_f64": "v_mul_f64", "v_fma_f64": "v_mul_f64", "v_cvt_f64_i32": "v_cvt_f64_i32", "v_cvt_f64_u32": "v_cvt_f64_i32",
    "v_ffbl_b32": "v_ffbl_b32", "v_ffbh_u32": "v_ffbl_b32", "v_lshl_add_u64": "v_lshlrev_b64",
}
DEFAULT_COST = 4.0


def assembly(path=None):
    """The device assembly of the units (concatenated); `path`: a directory to keep / reuse the .s files in."""
    from concurrent.futures import ThreadPoolExecutor
    d = path or "/tmp/pomdp_isa"
    os.makedirs(d, exist_ok=True)

    def one(u):
        out = os.path.join(d, u + ".s")
        src = os.path.join(CSRC, u)
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC)
                                                                  if f.endswith((".hip", ".h"))):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                   "--cuda-device-only", "-S", "-o", out, src], stderr=subprocess.DEVNULL)
        return open(out).read()

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        return "\n".join(ex.map(one, UNITS))


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def kernels_of(text):
    """-> {mangled name: [lines]}"""
    out, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                cur = None
                continue
            out[cur].append(line)
    return out


def blocks_of(lines):
    """Basic blocks with the compiler's loop annotations: [(label, header-of-depth or None, in-loop (header, depth) or None, [instructions])]"""
    blocks, cur = [], None
    i = 0
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        mb = re.match(r"^; %bb\.(\d+):(.*)", l)
        if m or mb:
            label = m.group(1) if m else "bb.%s" % mb.group(1)
            notes = [(m or mb).group(2)]
            j = i + 1
            while j < len(lines) and lines[j].strip().startswith(";") and not lines[j].startswith("; %bb."):
                notes.append(lines[j])
                j += 1
            note = " ".join(notes)
            hdr = re.search(r"Loop Header: Depth=(\d+)", note)
            inl = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
            par = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", note)
            cur = {"label": label, "header_depth": int(hdr.group(1)) if hdr else None,
                   "in_loop": (inl.group(1), int(inl.group(2))) if inl else None,
                   "parent": (par.group(1), int(par.group(2))) if par else None, "insts": []}
            blocks.append(cur)
            i = j
            continue
        s = l.strip()
        if cur is not None and s and not s.startswith((";", ".", "#")):
            cur["insts"].append(s.split()[0])
        i += 1
    return blocks


PHILOX_OPS = ("v_mad_u64_u32", "v_bitop3_b32")


def is_tie_path(prev, b):
    """An exec-masked block (the block before it ends in s_cbranch_execz, i.e. skips it when no lane takes it) that is
    essentially one Philox block: the low-word block of a draw whose high word left the comparison undecided (2^-27 per
    draw) — never on the common path, so it does not belong in the loop's mix."""
    if prev is None or not prev["insts"] or prev["insts"][-1] != "s_cbranch_execz":
        return False
    v = [i for i in b["insts"] if i.startswith("v_")]
    return len(v) >= 30 and sum(base(i) in PHILOX_OPS for i in v) >= 0.7 * len(v)


def hot_loop(blocks):
    """The loop (of any depth: the fused step loops sit inside the four-segment priority loop, LoopPrio) with the most
    instructions in its OWN blocks; the loops nested inside it and tie-path blocks left out.
    -> (loop header, instructions of the common path, VALU instructions left out as tie paths)"""
    loops, cold = {}, {}
    prev = None
    for b in blocks:
        key = None
        if b["header_depth"] is not None:
            key = b["label"].lstrip(".L")
        elif b["in_loop"]:
            key = b["in_loop"][0]
        if key:
            if is_tie_path(prev, b):
                cold[key] = cold.get(key, 0) + sum(i.startswith("v_") for i in b["insts"])
            else:
                loops.setdefault(key, []).extend(b["insts"])
        prev = b
    if not loops:
        return None, [], 0
    key = max(loops, key=lambda k: len(loops[k]))
    return key, loops[key], cold.get(key, 0)


def base(mn):
    return re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", mn)


def mix_of(insts, costs):
    valu = {}
    other = {"salu": 0, "lds": 0, "vmem": 0, "smem": 0}
    for mn in insts:
        if mn.startswith("v_"):
            b = base(mn)
            if mn.endswith("_dpp") and b == "v_mov_b32":
                b = "v_mov_b32_dpp"
            valu[b] = valu.get(b, 0) + 1
        elif mn.startswith("ds_"):
            other["lds"] += 1
        elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
            other["vmem"] += 1
        elif mn.startswith("s_load") or mn.startswith("s_buffer_load"):
            other["smem"] += 1
        elif mn.startswith("s_"):
            other["salu"] += 1
    n = sum(valu.values())
    cyc, unpriced = 0.0, {}
    for mn, c in valu.items():
        cls = "v_cmp_lt_u32" if mn.startswith("v_cmp") else ("v_mov_b32_dpp" if mn == "v_mov_b32_dpp" else ALIAS.get(mn))
        cost = costs.get(cls) if cls else None
        if cost is None:
            cost = DEFAULT_COST
            unpriced[mn] = c
        cyc += cost * c
    return {"valu_instructions": n, "cycles": cyc, "avg_cycles_per_instruction": cyc / n if n else None,
            "by_mnemonic": dict(sorted(valu.items(), key=lambda kv: -kv[1])), "unpriced_at_4_cycles": unpriced, "other": other}


def load_costs(path, w="W4"):
    """Issue cost per class: the measured figure with four waves per SIMD, rounded to the pipeline's own quantum — 2 cycles
    for the classes that measure below 3 (32 lanes per cycle: v_add / v_sub / v_and / v_or / v_xor / shifts / v_mov), else 4
    (16 lanes per cycle).  The measured figures carry the microbenchmark's loop overhead (2.2-2.3 and 4.2-4.4); rounding
    down keeps the derived peak an upper bound."""
    if not path:
        return {}
    d = json.load(open(path))
    return {k: (2.0 if v[w]["cycles"] < 3.0 else 4.0) for k, v in d["ops"].items()}


def main():
    args = sys.argv[1:]
    costs_path = out_json = asm = None
    pats = []
    while args:
        a = args.pop(0)
        if a == "--costs":
            costs_path = args.pop(0)
        elif a == "--json":
            out_json = args.pop(0)
        elif a == "--asm":
            asm = args.pop(0)
        else:
            pats.append(a)
    pats = pats or DEFAULT
    costs = load_costs(costs_path)
    ks = kernels_of(assembly(asm))
    names = demangle(list(ks))
    res = {"source": "tools/isa_mix.py: static mix of the hot loop (outermost loop with the most instructions, nested loops left out) of "
                     "hipcc --offload-arch=gfx950 -O3 -S gym_pomdp_amd/csrc/{fused_*,planner}.hip",
           "costs": costs_path and os.path.relpath(costs_path, REPO), "cost_column": "W4 (four waves per SIMD), rounded to 2 or 4 cycles", "kernels": {}}
    for pat in pats:
        for mangled, lines in ks.items():
            nm = names[mangled]
            if pat not in nm:
                continue
            short = nm.replace("void pomdp::", "").replace("pomdp::", "").split("(")[0]
            key, insts, cold = hot_loop(blocks_of(lines))
            if not insts:
                continue
            m = mix_of(insts, costs)
            m["loop"] = key
            m["tie_path_valu_left_out"] = cold
            res["kernels"][short] = m
            top = ", ".join("%s %d" % kv for kv in list(m["by_mnemonic"].items())[:8])
            print("%-70s loop %-10s VALU %4d (+%d tie-path)  avg %.2f cycles  (salu %d, lds %d, vmem %d)  %s" % (
                short[:70], key, m["valu_instructions"], cold, m["avg_cycles_per_instruction"] or 0, m["other"]["salu"], m["other"]["lds"],
                m["other"]["vmem"], top))
            if m["unpriced_at_4_cycles"]:
                print("    unpriced (4 cycles assumed): %s" % m["unpriced_at_4_cycles"])
    if out_json:
        sys.path.insert(0, REPO)
        from bench import csrc_sha                      # the sources this mix was compiled from (bench.py: counters_stale)
        res["csrc_sha256"] = csrc_sha()
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
