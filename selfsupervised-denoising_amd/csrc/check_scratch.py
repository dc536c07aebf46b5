"""Build check (csrc/Makefile): the weight-gradient kernels must not use scratch memory.  Their bodies keep up to 21 32x32 fp32
accumulators per wave in registers (480+ of the 512); when the allocator tips over, the spills land inside the K loops AND the runtime
throttles a dispatch with that much scratch per lane to a fraction of the CUs (measured: 2.2 ms instead of 0.45 ms).
usage: check_scratch.py <object file> <kernel name prefix> ..."""
import re
import subprocess
import sys
import tempfile
import os

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def main():
    # optional: --sgpr-budget N (spilled SGPRs: v_readlane / v_writelane on the VALU port the MFMAs issue from), --only REGEX (mangled names)
    args = sys.argv[1:]
    budget, only = None, None
    while args and args[0].startswith("--"):
        if args[0] == "--sgpr-budget":
            budget = int(args[1])
        elif args[0] == "--only":
            only = re.compile(args[1])
        else:
            sys.exit("check_scratch: unknown option " + args[0])
        args = args[2:]
    obj, prefixes = args[0], args[1:]
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), tmp)
        subprocess.run([OBJDUMP, "--offloading", tmp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [os.path.join(d, f) for f in os.listdir(d) if "amdgcn" in f]
        assert cos, "no device code object in " + obj
        notes = subprocess.run([READELF, "--notes", cos[0]], check=True, capture_output=True, text=True).stdout
    bad, seen = [], 0
    for blk in notes.split("- .agpr_count")[1:]:
        m = re.search(r"\.name:\s+(\S+)", blk)
        if not m or not any(p in m.group(1) for p in prefixes) or (only and not only.search(m.group(1))):
            continue
        seen += 1
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        vspill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        if scratch or vspill:
            bad.append("%s: %d B of scratch per lane, %d spilled VGPRs" % (m.group(1), scratch, vspill))
        sspill = int(re.search(r"\.sgpr_spill_count:\s+(\d+)", blk).group(1))
        if budget is not None and sspill > budget:
            bad.append("%s: %d spilled SGPRs (budget %d)" % (m.group(1), sspill, budget))
    assert seen, "no kernel matching %s in %s" % (prefixes, obj)
    if bad:
        sys.exit("check_scratch: " + "; ".join(bad))
    print("check_scratch: %d kernels of %s without scratch" % (seen, os.path.basename(obj)))


if __name__ == "__main__":
    main()
