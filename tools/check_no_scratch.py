"""Build-time check: every kernel of the given HIP objects must run without scratch — the kernels named here keep loads
in flight behind hand-counted s_waitcnt vmcnt(N) (inline asm the compiler does not see); a register the compiler spills or copies
between such a load and its wait would silently hold garbage.  usage: python tools/check_no_scratch.py obj.o [obj.o ...]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
bad = 0
for obj in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), tmp)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", tmp], cwd=d, check=True, capture_output=True)
        cos = [f for f in os.listdir(d) if "amdgcn" in f]
        assert cos, "no device code object in %s" % obj
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(d, cos[0])], check=True, capture_output=True, text=True).stdout
    n = 0
    for blk in notes.split(".name:")[1:]:
        name = blk.split()[0]
        m = re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk)
        s = re.search(r"\.vgpr_spill_count:\s*(\d+)", blk)
        if m is None:
            continue
        n += 1
        if int(m.group(1)) != 0 or (s and int(s.group(1)) != 0):
            print("%s: kernel %s needs scratch (private_segment_fixed_size %s, vgpr_spill_count %s)" % (obj, name, m.group(1), s.group(1) if s else "?"))
            bad += 1
    print("%s: %d kernels, %s" % (obj, n, "no scratch" if not bad else "SCRATCH"))
sys.exit(1 if bad else 0)
