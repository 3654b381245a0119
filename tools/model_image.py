"""Reader for the flat device model image (`*.img`) written by
`oracle/_ref/ref_dump export` and consumed by jumanpp_amd/csrc/model_image.cc.

Layout: 8-byte magic "JPPGPUI1", then 8-byte-aligned sections
  u32 tag, u32 aux, u64 size, payload[size], pad to 8
terminated by an all-zero header.
"""
import struct

SEC_INFO, SEC_TRIE, SEC_ENTRY_PTRS, SEC_ENTRY_DATA, SEC_WEIGHTS, SEC_UNK, \
    SEC_FEATURES, SEC_FIELDS, SEC_STRINGS, SEC_INTS, SEC_RNN = range(1, 12)


def read_sections(path):
    data = open(path, 'rb').read()
    assert data[:8] == b'JPPGPUI1', 'bad magic'
    pos = 8
    out = []
    while True:
        pos = (pos + 7) & ~7
        tag, aux, size = struct.unpack_from('<IIQ', data, pos)
        pos += 16
        if tag == 0:
            break
        out.append((tag, aux, data[pos:pos + size]))
        pos += size
    return out


class IntReader:
    def __init__(self, buf):
        self.buf = buf
        self.pos = 0

    def i32(self):
        v, = struct.unpack_from('<i', self.buf, self.pos)
        self.pos += 4
        return v

    def ints(self):
        n = self.i32()
        return [self.i32() for _ in range(n)]


def parse_features(buf):
    r = IntReader(buf)
    prims = []
    for _ in range(r.i32()):
        kind = r.i32()
        prims.append((kind, r.ints()))
    comps = []
    for _ in range(r.i32()):
        prim = r.i32()
        comps.append((prim, r.ints(), r.ints()))
    pats = []
    for _ in range(r.i32()):
        idx = r.i32()
        pats.append((idx, r.ints()))
    ngrams = []
    for _ in range(r.i32()):
        idx = r.i32()
        ngrams.append((idx, r.ints()))
    return dict(prims=prims, comps=comps, pats=pats, ngrams=ngrams)


def parse_info(buf):
    names = ['num_features', 'num_data', 'num_placeholders', 'entry_count',
             'num_patterns', 'num_uni_only', 'num_string_storages', 'num_int_storages']
    return dict(zip(names, struct.unpack_from('<8i', buf, 0)))
