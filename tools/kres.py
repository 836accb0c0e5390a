#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of the product library, from the assembly (no GPU needed).

  python tools/kres.py [git-rev]      # default: the working tree; with a rev, that revision's csrc (checked out under /tmp)
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def census(src_dir):
    asm = "/tmp/lfvio_kres.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-I", os.path.join(ROOT, "include"), os.path.join(src_dir, "lfvio_hip.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    out, name = {}, None
    for ln in open(asm):
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", ln)
        if m:
            name = subprocess.check_output(["c++filt", m.group(1)], text=True).strip()
            name = re.sub(r"\(.*", "", name)
            out[name] = {}
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|group_segment_fixed_size|private_segment_fixed_size)\s+(\d+)", ln)
        if m and name:
            out[name][m.group(1)] = int(m.group(2))
    return out


if __name__ == "__main__":
    src = os.path.join(ROOT, "lf-vio_amd", "csrc")
    if len(sys.argv) > 1:
        tmp = "/tmp/kres_rev"
        subprocess.check_call(f"rm -rf {tmp} && mkdir -p {tmp} && git -C {ROOT} archive {sys.argv[1]} lf-vio_amd/csrc include | tar -x -C {tmp}", shell=True)
        src = os.path.join(tmp, "lf-vio_amd", "csrc")
    for k, v in sorted(census(src).items()):
        print(f"{k:70s} vgpr {v.get('next_free_vgpr', 0):4d} (arch {v.get('accum_offset', 0):3d}) sgpr {v.get('next_free_sgpr', 0):4d} "
              f"lds {v.get('group_segment_fixed_size', 0):7d} scratch {v.get('private_segment_fixed_size', 0):5d}")
