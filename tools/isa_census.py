#!/usr/bin/env python3
"""Static instruction census of the gfx950 code of selected kernels (runs anywhere hipcc does: no GPU needed).

  python tools/isa_census.py [out.md] [kernel-name-substring ...]

Compiles lf-vio_amd/csrc/lfvio_hip.hip to assembly (--cuda-device-only -S), cuts out every kernel whose demangled name contains
one of the substrings (default: the role instantiations of k_lin and the window-resident kernels) and counts its instructions
by class.  Static counts: a loop body counts once — the k_lin roles are compiled as kernels of their own (k_lin<1> landmark
role, k_lin<2> Gram role, k_lin<8> pose-side roles), so the table is per role; the dynamic totals of the same kernels
(SQ_INSTS_VALU, SQ_INSTS_LDS, ... per launch) are in profiles/rNN/pmc_summary.md."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lf-vio_amd", "csrc", "lfvio_hip.hip")
CLASSES = [
    ("FP64 arithmetic (v_fma / v_mul / v_add / v_fmac / min / max _f64)", r"^v_(fma|fmac|mul|add|min|max|ldexp|frexp\w*|trunc|floor|rndne|cvt)_?\w*f64"),
    ("FP64 transcendental (v_rcp / v_rsq / v_sqrt / v_div*)", r"^v_(rcp|rsq|sqrt|div_\w+)_f64"),
    ("matrix pipe (v_mfma_f64)", r"^v_mfma"),
    ("select / compare (v_cndmask, v_cmp*)", r"^v_(cndmask|cmp)"),
    ("cross-lane (DPP moves, v_readlane / v_writelane / readfirstlane, permlane, ds_bpermute)", r"^(v_readlane|v_writelane|v_readfirstlane|v_permlane|ds_bpermute|ds_permute|v_mov_b\d+_dpp|\w+_dpp)"),
    ("integer / address VALU (v_add_u32, v_lshl*, v_mad_u*, v_mul_lo/hi, v_and/or, v_ashr, v_bfe ...)", r"^v_(add|sub|subrev|lshl|lshr|ashr|mad|mul_lo|mul_hi|mul_u|mul_i|and|or|xor|not|bfe|bfi|alignbit|add3|lshl_add|add_lshl|addc|subb|min_[ui]|max_[ui]|ffb|bcnt|mbcnt|perm)"),
    ("register moves (v_mov, v_accvgpr*)", r"^v_(mov|accvgpr|swap)"),
    ("LDS (ds_read / ds_write / ds_add)", r"^ds_"),
    ("vector memory loads (global_load / buffer_load / scratch_load)", r"^(global_load|buffer_load|flat_load|scratch_load)"),
    ("vector memory stores / atomics", r"^(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic)"),
    ("scalar memory (s_load, s_buffer_load)", r"^s_(load|buffer_load|store|dcache)"),
    ("waits (s_waitcnt, s_nop, s_barrier, s_sleep)", r"^s_(waitcnt|nop|barrier|sleep)"),
    ("branches (s_cbranch, s_branch, exec mask handling)", r"^s_(cbranch|branch|and_saveexec|or_saveexec|andn2_saveexec|mov_b64 exec|or_b64 exec|xor_b64 exec|andn2_b64 exec|and_b64 exec|setpc|swappc|getpc|endpgm)"),
    ("scalar ALU (s_add, s_mul, s_lshl, s_cmp, s_mov, s_cselect ...)", r"^s_"),
]


def main():
    args = sys.argv[1:]
    out = args[0] if args and args[0].endswith(".md") else None
    subs = [a for a in args if not a.endswith(".md")] or ["k_lin<", "k_linw", "k_stepw", "k_solve_dense"]
    asm = "/tmp/lfvio_census.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only", SRC, "-o", asm],
                          stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()
    kernels, cur, name = {}, None, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end") or ".amdhsa_kernel" in ln:
                kernels[name], cur = cur, None
            else:
                s = ln.strip()
                if s and not s.startswith((";", ".", "//")) and not s.endswith(":"):
                    cur.append(s.split()[0] + (" exec" if " exec" in s and s.startswith("s_") else ""))
    dem = subprocess.run(["c++filt"] + list(kernels), capture_output=True, text=True).stdout.splitlines()
    pick = [(d, kernels[k]) for k, d in zip(kernels, dem) if any(sub in d for sub in subs) and d.split("(")[0].strip().startswith(("void k_", "k_"))]
    md = ["# static instruction census of the gfx950 code (tools/isa_census.py; counts per kernel body, a loop body counts once)", ""]
    md += ["| class | " + " | ".join(d.split("(")[0].replace("void ", "") for d, _ in pick) + " |", "|---|" + "---|" * len(pick)]
    table = []
    for d, ins in pick:
        c = collections.Counter()
        for op in ins:
            for label, rx in CLASSES:
                if re.match(rx, op):
                    c[label] += 1
                    break
            else:
                c["other"] += 1
        table.append((c, len(ins)))
    for label, _ in CLASSES + [("other", None)]:
        md.append(f"| {label} | " + " | ".join(f"{c[label]} ({100.0 * c[label] / n:.0f} %)" for c, n in table) + " |")
    md.append("| total | " + " | ".join(str(n) for _, n in table) + " |")
    text = "\n".join(md)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
