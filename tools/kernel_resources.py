#!/usr/bin/env python
"""Per-kernel register / LDS / scratch usage of one csrc/*.hip file (device-only compile, no GPU needed).

    python tools/kernel_resources.py conv_mfma.hip [substring-filter] [--reuse]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
path = src if os.path.exists(src) else os.path.join(ROOT, "m-cedm_amd", "csrc", src)
co = os.path.join("/tmp/regs", os.path.basename(path) + ".co")
os.makedirs("/tmp/regs", exist_ok=True)
flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/m-cedm_amd/csrc"]
if os.path.basename(path) in ("edm.hip", "pde.hip"):
    flags.append("-ffp-contract=off")
flags += os.environ.get("MCEDM_EXTRA_HIPCC_FLAGS", "").split()
if "--reuse" not in sys.argv or not os.path.exists(co):
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-c", path, "-o", co], check=True)
elf = co + ".elf"
if "--reuse" not in sys.argv or not os.path.exists(elf):
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", f"--input={co}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"], check=True)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
for blk in notes.split("- .agpr_count:")[1:]:
    get = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
    name = demangle(get("name"))
    if flt in name:
        print(f"{name[:110]:110s} vgpr {get('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {get('sgpr_count'):>4s} "
              f"spill {get('vgpr_spill_count'):>3s} scratch {get('private_segment_fixed_size'):>5s} lds {get('group_segment_fixed_size'):>6s}")
