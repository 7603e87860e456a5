"""VGPR / SGPR / LDS / spill figures of the kernels in libmdt_hip.so whose (mangled) name contains a pattern, read from the
code objects' notes (no GPU needed).  usage: python tools/kernel_resources.py <pattern> [library]"""
import os, re, shutil, subprocess, sys, tempfile
pat = sys.argv[1]
lib = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "mdt_policy_amd", "csrc", "libmdt_hip.so"))
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    shutil.copy(lib, os.path.join(d, "lib.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=d, capture_output=True)  # writes lib.so.N.<target> beside it
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], cwd=d, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name or pat not in name.group(1):
                continue
            g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [0, "?"])[1]
            filt = shutil.which("c++filt")
            dem = subprocess.run([filt, name.group(1)], capture_output=True, text=True).stdout.strip() if filt else name.group(1)
            print(f"{dem[:90]:90s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} lds {g('group_segment_fixed_size'):>6} "
                  f"scratch {g('private_segment_fixed_size'):>4} vspill {g('vgpr_spill_count')} sspill {g('sgpr_spill_count')}")
