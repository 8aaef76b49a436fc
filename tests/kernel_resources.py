"""prints VGPR / SGPR / LDS / scratch / spill figures of the kernels in the built gfx950 library whose name contains any of the
given substrings (code-object metadata; no GPU needed).   python tests/kernel_resources.py conv3 modw"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path('/opt/rocm/lib/llvm/bin')


def main():
    pats = sys.argv[1:] or ['']
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = f'{tmp}/fat.bin', f'{tmp}/gfx950.co'
        subprocess.check_call([str(LLVM / 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}',
                               str(ROOT / 'gigagan_pytorch_amd' / 'libgigagan_amd.so'), f'{tmp}/discard.so'])
        subprocess.check_call([str(LLVM / 'clang-offload-bundler'), '--unbundle', '--type=o',
                               '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--input={fat}', f'--output={co}'])
        notes = subprocess.check_output([str(LLVM / 'llvm-readelf'), '--notes', co], text=True)
    for k in notes.split('- .agpr_count')[1:]:
        name = re.search(r'\.name:\s+(\S+)', k).group(1)
        if not any(p in name for p in pats):
            continue
        g = lambda f: re.search(r'\.' + f + r':\s+(\d+)', k).group(1)
        print(f"{name[:100]:100s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} "
              f"scratch {g('private_segment_fixed_size'):>5s} spill {g('vgpr_spill_count')}")


if __name__ == '__main__':
    main()
