"""Register / LDS / scratch budget of every kernel in libttenv.so, read from the code objects inside the library (no GPU, no ROCm
tool: the clang offload bundles and the AMDGPU metadata note are parsed here).

    python -m ddpg_trucktrailer_amd.kernel_resources            -> a table
    kernels()                                                   -> {demangled-ish name: {...}}

Why it matters: beside the policy's 171 workgroups 85 CUs are free, and a weight-gradient launch has 200-210
workgroups -- they start together only if three fit a CU, i.e. <= 168 registers; the policy kernel must not spill (a spilled
build ran at 0.144 ms per step instead of 0.085); k_step needs <= 128 to keep four waves per SIMD.  tests/test_kernel_resources.py
holds those budgets."""
import os
import struct

import msgpack

from . import _lib

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob, arch="gfx950"):
    pos = 0
    while True:
        i = blob.find(_MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode()
            off += tl
            if size and arch in triple:
                yield blob[i + o:i + o + size]
        pos = i + 24


def _notes(elf):
    """Descriptors of the AMDGPU metadata notes (NT_AMDGPU_METADATA = 32) of an ELF64 little-endian code object."""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2 and elf[5] == 1
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = shoff + k * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:      # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b"AMDGPU" and ntype == 32:
                yield desc


def kernels(path=None):
    """{kernel symbol: {"vgpr": total registers per lane (VGPR + AGPR, as allocated), "sgpr", "lds", "scratch", "vgpr_spills",
    "sgpr_spills", "max_threads"}} for every gfx950 kernel in the library."""
    blob = open(path or _lib.LIB_PATH, "rb").read()
    out = {}
    for co in _code_objects(blob):
        for desc in _notes(co):
            meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in meta.get("amdhsa.kernels", []):
                out[k[".name"]] = {
                    "vgpr": int(k.get(".vgpr_count", 0)), "agpr": int(k.get(".agpr_count", 0)), "sgpr": int(k.get(".sgpr_count", 0)),
                    "lds": int(k.get(".group_segment_fixed_size", 0)), "scratch": int(k.get(".private_segment_fixed_size", 0)),
                    "vgpr_spills": int(k.get(".vgpr_spill_count", 0)), "sgpr_spills": int(k.get(".sgpr_spill_count", 0)),
                    "max_threads": int(k.get(".max_flat_workgroup_size", 0)),
                    "kernarg_preload": int(k.get(".kernarg_segment_size", 0)),
                }
    return out


def waves_per_simd(vgpr):
    """Waves of a kernel that fit one SIMD's 512 registers per lane (allocation granule 8)."""
    alloc = max(8, (vgpr + 7) // 8 * 8)
    return min(8, 512 // alloc)


def find(ks, *parts):
    """The kernels whose mangled name contains every given part."""
    return {n: v for n, v in ks.items() if all(p in n for p in parts)}


if __name__ == "__main__":
    ks = kernels()
    print(f"{os.path.basename(_lib.LIB_PATH)}: {len(ks)} kernels")
    print(f"{'registers':>9} {'waves/SIMD':>10} {'sgpr':>5} {'LDS B':>7} {'scratch B':>9} {'vgpr spills':>11}  kernel")
    for n, v in sorted(ks.items(), key=lambda kv: -kv[1]["vgpr"]):
        print(f"{v['vgpr']:9d} {waves_per_simd(v['vgpr']):10d} {v['sgpr']:5d} {v['lds']:7d} {v['scratch']:9d} {v['vgpr_spills']:11d}  {n[:110]}")
