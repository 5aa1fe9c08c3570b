"""Disassemble the gfx950 code object of a build/obj/*.o:
python scripts/disasm.py build/obj/pm_conv_bf16.o out.s"""
import subprocess
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
from check_spills import code_objects

blob = next(code_objects(Path(sys.argv[1])))
Path('/tmp/_pm.co').write_bytes(blob)
text = subprocess.run(
    ['/opt/rocm/lib/llvm/bin/llvm-objdump', '-d', '--demangle', '/tmp/_pm.co'],
    capture_output=True, text=True, check=True).stdout
Path(sys.argv[2]).write_text(text)
