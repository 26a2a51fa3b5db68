#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of a built object or .so (gfx950 code objects inside).
Usage: tools/kernel_resources.py [paths...] [--filter SUBSTR]   (default: csrc/build/*.o)"""
import glob, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"

def code_objects(path, tmp):
    out = os.path.join(tmp, os.path.basename(path) + ".co")
    fat = os.path.join(tmp, os.path.basename(path) + ".fat")
    if subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path], capture_output=True).returncode:
        return None
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                        f"--output={out}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
    return out if r.returncode == 0 and os.path.getsize(out) > 0 else None

def kernels(co):
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    recs = []
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        recs.append(dict(name=dem, vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"),
                         spill=g("vgpr_spill_count"), lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size")))
    return recs

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = None
    if "--filter" in sys.argv: flt = sys.argv[sys.argv.index("--filter") + 1]; args = [a for a in args if a != flt]
    here = os.path.dirname(os.path.abspath(__file__))
    paths = args or sorted(glob.glob(os.path.join(here, "..", "oct-image-segmentation-models_amd", "csrc", "build", "*.o")))
    with tempfile.TemporaryDirectory() as tmp:
        for p in paths:
            co = code_objects(p, tmp)
            if not co: continue
            for k in kernels(co):
                if flt and flt not in k["name"]: continue
                nm = re.sub(r"^void oct::", "", k["name"]).replace("(oct::IgemmArgs)", "").replace("(oct::ConvBwdWArgs)", "")
                print(f"{k['vgpr']:>4} v {k['agpr']:>3} a {k['sgpr']:>3} s  spill {k['spill']:>3}  scratch {k['scratch']:>5}  lds {k['lds']:>6}  {nm}")

if __name__ == "__main__":
    main()
