"""`.bin` container of the reference API (cra5/api/cra5_api.py:108-117, 132-140 with the
big-endian helpers of cra5/api/utils.py:10-34): uint32 zH, zW, n_strings (=2), then per
string uint32 length + raw bytes, order [y, z]; batch item 0 only."""
import struct


def write_uints(fd, values):
    fd.write(struct.pack(">%dI" % len(values), *values))
    return len(values) * 4


def write_bytes(fd, values):
    if len(values) == 0:
        return 0  # (the reference returns None here, which its caller then fails to add)
    fd.write(values)
    return len(values)


def read_uints(fd, n):
    return struct.unpack(">%dI" % n, fd.read(4 * n))


def read_bytes(fd, n):
    return fd.read(n)


def pack_bin(strings, z_shape):
    """strings: [[y_bytes, ...], [z_bytes, ...]] as returned by compress() -> bytes."""
    out = [struct.pack(">3I", int(z_shape[0]), int(z_shape[1]), len(strings))]
    for s in strings:
        out.append(struct.pack(">I", len(s[0])))
        out.append(s[0])
    return b"".join(out)


def write_bin(f, strings, z_shape):
    """pack_bin straight into an open binary file: the 4.5 MB y stream is handed to the file once instead of being
    copied into a joined bytes object first."""
    f.write(struct.pack(">3I", int(z_shape[0]), int(z_shape[1]), len(strings)))
    for s in strings:
        f.write(struct.pack(">I", len(s[0])))
        f.write(s[0])


def unpack_bin(data):
    """bytes -> (strings [[y],[z]], z_shape)."""
    zh, zw, n = struct.unpack(">3I", data[:12])
    off = 12
    strings = []
    for _ in range(n):
        (ln,) = struct.unpack(">I", data[off:off + 4])
        off += 4
        strings.append([bytes(data[off:off + ln])])
        off += ln
    return strings, (zh, zw)
