"""Build check: no kernel of libpromonet_hip.so may spill registers.

Extracts the gfx950 code object of every build/obj/*.o (clang offload
bundle), reads the kernel metadata with llvm-readelf and fails when a kernel
has .vgpr_spill_count != 0 or private (scratch) memory.
`make` runs it after linking; `python scripts/check_spills.py -v` lists every
kernel's registers / LDS.
"""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
# Kernels allowed to use scratch (none). Kept explicit so that an exception
# is a reviewed decision, not a silent regression.
ALLOWED = set()


def code_objects(path):
    data = path.read_bytes()
    start = data.find(MAGIC)
    while start >= 0:
        count = struct.unpack_from('<Q', data, start + len(MAGIC))[0]
        cursor = start + len(MAGIC) + 8
        for _ in range(count):
            offset, size, length = struct.unpack_from('<QQQ', data, cursor)
            cursor += 24
            triple = data[cursor:cursor + length].decode()
            cursor += length
            if 'gfx950' in triple and size:
                yield data[start + offset:start + offset + size]
        start = data.find(MAGIC, start + 1)


def kernels(blob):
    with tempfile.NamedTemporaryFile(suffix='.co') as file:
        file.write(blob)
        file.flush()
        notes = subprocess.run(
            [READELF, '--notes', file.name], capture_output=True, text=True,
            check=True).stdout
    for block in notes.split('- .agpr_count:')[1:]:
        fields = dict(re.findall(r'\.(\w+):\s+(\S+)', '.agpr_count:' + block))
        if 'name' in fields:
            yield fields


def main():
    verbose = '-v' in sys.argv
    bad = []
    seen = 0
    objects = [Path(a) for a in sys.argv[1:] if not a.startswith('-')] or \
        sorted((ROOT / 'build' / 'obj').glob('pm_*.o'))
    for obj in objects:
        for blob in code_objects(obj):
            for k in kernels(blob):
                seen += 1
                # (SGPR spills go to VGPR lanes, not memory: reported with
                # -v, not an error)
                spills = int(k.get('vgpr_spill_count', 0))
                scratch = int(k.get('private_segment_fixed_size', 0))
                name = subprocess.run(
                    ['c++filt', k['name']], capture_output=True,
                    text=True).stdout.strip()[:110]
                if verbose:
                    print(f"{k.get('vgpr_count'):>4} vgpr {k.get('agpr_count'):>3} agpr "
                          f"{k.get('sgpr_count'):>3} sgpr {k.get('group_segment_fixed_size'):>6} lds "
                          f"{spills} spill {scratch} scratch  {name}")
                # a scratch segment with no VGPR spill and some SGPR spills is
                # the spill-slot bookkeeping of SGPRs that went to VGPR lanes:
                # no scratch instruction is emitted for it
                sgpr_only = not spills and int(k.get('sgpr_spill_count', 0))
                if (spills or (scratch and not sgpr_only)) and \
                        k['name'] not in ALLOWED:
                    bad.append((name, spills, scratch))
    if not seen:
        sys.exit('check_spills: no kernels found (build first)')
    for name, spills, scratch in bad:
        print(f'SPILL: {name}: {spills} spilled registers, '
              f'{scratch} B scratch', file=sys.stderr)
    if bad:
        sys.exit(f'check_spills: {len(bad)} of {seen} kernels use scratch')
    print(f'check_spills: {seen} kernels, no spills, no scratch')


if __name__ == '__main__':
    main()
