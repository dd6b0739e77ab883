"""The device code of a built library, as one number: sha256 of the `.hip_fatbin` section of librcgpu.so (every kernel of every .hip file,
as hipcc laid them into the shared object).  profiles/traffic.json records it beside the sha256 of the kernel SOURCES the PMC passes ran on:
an edit to the host code inside a .hip file changes the source and not the kernels, and the measured bytes still describe them."""
import hashlib
import struct


def fatbin_sha256(path):
    """sha256 hex of the .hip_fatbin section of the ELF64 file at `path`; None when there is none (or no such file)."""
    try:
        b = open(path, "rb").read()
        if b[:4] != b"\x7fELF" or b[4] != 2:
            return None
        shoff, = struct.unpack_from("<Q", b, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
        sec = [struct.unpack_from("<IIQQQQ", b, shoff + i * shentsize) for i in range(shnum)]      # name, type, flags, addr, offset, size
        stroff = sec[shstrndx][4]
        for name, _, _, _, off, size in sec:
            end = b.index(b"\0", stroff + name)
            if b[stroff + name:end] == b".hip_fatbin":
                return hashlib.sha256(b[off:off + size]).hexdigest()
    except Exception:
        pass
    return None
