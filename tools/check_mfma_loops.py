"""Build-time guard for the match kernels (match.hip, match16.hip): their MFMAs are inline asm, so the compiler's
hazard recogniser does not see them - a spill / reload / copy of an accumulator next to an MFMA would read it
before the matrix pipe has written it.  The kernels are written so that no accumulator is spilled inside the
MFMA loop; this script compiles them to gfx950 assembly and FAILS if any kernel has a scratch instruction between
its first and last v_mfma (also a performance bug: an accumulator going through memory every k-step).

    python tools/check_mfma_loops.py [file.hip ...]      (default: both match kernels)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kikuchipy_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernels_with_spills_in_mfma_loop(hip_file):
    """{kernel: [offending lines]} for the kernels of `hip_file` (a path under csrc/)."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                        "--cuda-device-only", os.path.join(CSRC, hip_file), "-o", out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    bad, checked = {}, 0
    name, body = None, []
    for line in text + ["\t.end"]:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m or line.strip() == ".end":
            if name and body:
                idx = [i for i, l in enumerate(body) if "v_mfma" in l]
                if idx:
                    checked += 1
                    off = [l.strip() for l in body[idx[0]:idx[-1] + 1] if re.search(r"\bscratch_(load|store)", l)]
                    if off:
                        bad[name] = off
            name, body = (m.group(1) if m else None), []
        elif name:
            body.append(line)
            if "s_endpgm" in line:
                pass
    return bad, checked


def main(files):
    rc = 0
    for f in files:
        bad, checked = kernels_with_spills_in_mfma_loop(f)
        print(f"{f}: {checked} kernels with MFMA loops checked, {len(bad)} with scratch traffic inside the loop")
        for k, lines in bad.items():
            print(f"  {k}: {len(lines)} scratch instructions, e.g. {lines[0]}")
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or ["match.hip", "match16.hip"]))
