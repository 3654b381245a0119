"""ctypes binding of the C ABI in include/jppgpu.h.

The product library is jumanpp_amd/libjppgpu.so (hipcc, gfx950).  There is no
CPU implementation behind this module: creating a context without a HIP device
raises.  (tests/ may point `lib_path` at the emulator build of the same kernel
sources; that is test infrastructure.)
"""
import ctypes as C
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, 'libjppgpu.so')


class UnkMaker(C.Structure):
    _fields_ = [('type', C.c_int32), ('char_class', C.c_int32), ('pattern_ptr', C.c_int32),
                ('priority', C.c_int32), ('placeholder', C.c_int32), ('replace_mask', C.c_uint32)]


class Model(C.Structure):
    _fields_ = [('trie', C.c_void_p), ('trie_bytes', C.c_size_t),
                ('entry_ptrs', C.c_void_p), ('entry_ptrs_bytes', C.c_size_t),
                ('entry_data', C.c_void_p), ('entry_data_bytes', C.c_size_t),
                ('weights', C.c_void_p), ('weight_exponent', C.c_uint32),
                ('num_features', C.c_int32), ('num_placeholders', C.c_int32),
                ('unk_makers', C.POINTER(UnkMaker)), ('num_unk_makers', C.c_int32),
                ('feature_spec', C.c_void_p), ('feature_spec_bytes', C.c_size_t),
                ('has_rnn', C.c_int32),
                ('rnn_known_index', C.c_void_p), ('rnn_known_index_bytes', C.c_size_t),
                ('rnn_unk_index', C.c_void_p), ('rnn_unk_index_bytes', C.c_size_t),
                ('rnn_matrix', C.c_void_p), ('rnn_embeddings', C.c_void_p), ('rnn_nce_embeddings', C.c_void_p),
                ('rnn_maxent', C.c_void_p),
                ('rnn_layer_size', C.c_uint32), ('rnn_maxent_order', C.c_uint32),
                ('rnn_maxent_size', C.c_uint64), ('rnn_vocab_size', C.c_uint64),
                ('rnn_nce_constant', C.c_float), ('rnn_unk_id', C.c_int32),
                ('rnn_unk_constant', C.c_float), ('rnn_unk_length', C.c_float),
                ('rnn_num_fields', C.c_uint32), ('rnn_fields', C.c_uint32 * 8)]


class FieldStorage(C.Structure):
    """jppgpu_field_storage"""
    _fields_ = [('column', C.c_int32), ('kind', C.c_int32), ('align_power', C.c_uint32), ('reserved', C.c_uint32),
                ('data', C.c_void_p), ('bytes', C.c_uint64)]


class Config(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('beam', C.c_int32), ('global_beam', C.c_int32), ('right_check', C.c_int32),
                ('right_beam', C.c_int32), ('max_input_bytes', C.c_int32), ('device', C.c_int32),
                ('use_rnn', C.c_int32), ('weight_perceptron', C.c_float), ('weight_rnn', C.c_float),
                ('dynamic_features', C.c_int32), ('num_host_scorers', C.c_int32), ('weight_host', C.c_float * 2),
                ('t0_memo_image', C.c_void_p), ('t0_memo_image_bytes', C.c_uint64), ('t0_memo_slots', C.c_uint32),
                ('keep_t0_memo_image', C.c_int32),
                ('field_storages', C.POINTER(FieldStorage)), ('num_field_storages', C.c_uint32), ('reserved1', C.c_uint32)]


class ResultView(C.Structure):
    _fields_ = [('n_sentences', C.c_uint32),
                ('status', C.c_void_p), ('n_codepoints', C.c_void_p), ('n_nodes', C.c_void_p),
                ('node_base', C.c_void_p), ('bnd_base', C.c_void_p),
                ('total_nodes', C.c_uint64), ('total_boundaries', C.c_uint64),
                ('beam', C.c_int32), ('global_beam', C.c_int32), ('num_scorers', C.c_int32), ('entry_row_stride', C.c_int32),
                ('path_len', C.c_void_p), ('path_nodes', C.c_void_p),
                ('nodes', C.c_void_p), ('unk', C.c_void_p),
                ('bnd_first', C.c_void_p), ('bnd_count', C.c_void_p),
                ('end_first', C.c_void_p), ('end_count', C.c_void_p), ('end_nodes', C.c_void_p),
                ('entry_rows', C.c_void_p), ('patterns', C.c_void_p), ('t0_scores', C.c_void_p),
                ('beams', C.c_void_p), ('cells', C.c_void_p), ('kept', C.c_void_p),
                ('gbeam_count', C.c_void_p), ('gbeam', C.c_void_p)]


class NbestView(C.Structure):
    _fields_ = [('n_sentences', C.c_uint32), ('n_best', C.c_int32), ('beam', C.c_int32), ('global_beam', C.c_int32),
                ('num_scorers', C.c_int32),
                ('status', C.c_void_p), ('n_codepoints', C.c_void_p), ('n_nodes', C.c_void_p),
                ('eos', C.c_void_p), ('path_first', C.c_void_p), ('items', C.c_void_p)]


class NgramsView(C.Structure):
    _fields_ = [('n_sentences', C.c_uint32), ('n_ngram', C.c_uint32), ('path_first', C.c_void_p), ('path_nodes', C.c_void_p),
                ('features', C.c_void_p)]


class SeedView(C.Structure):
    _fields_ = [('n_sentences', C.c_uint32), ('status', C.c_void_p), ('n_codepoints', C.c_void_p), ('n_seeds', C.c_void_p),
                ('seed_base', C.c_void_p), ('seeds', C.c_void_p), ('unk', C.c_void_p)]


class ExtraSeeds(C.Structure):
    _fields_ = [('offsets', C.c_void_p), ('seeds', C.c_void_p)]


SEED_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(SeedView), C.POINTER(ExtraSeeds))


class LatticePairs(C.Structure):
    _fields_ = [('n_sentences', C.c_uint32), ('num_features', C.c_int32), ('status', C.c_void_p), ('n_codepoints', C.c_void_p),
                ('n_nodes', C.c_void_p), ('node_base', C.c_void_p), ('total_nodes', C.c_uint64), ('nodes', C.c_void_p),
                ('unk', C.c_void_p), ('entry_rows', C.c_void_p), ('bnd_base', C.c_void_p), ('total_boundaries', C.c_uint64),
                ('bnd_first', C.c_void_p), ('bnd_count', C.c_void_p), ('end_first', C.c_void_p), ('end_count', C.c_void_p),
                ('end_nodes', C.c_void_p), ('pair_base', C.c_void_p), ('total_pairs', C.c_uint64)]


SCORE_LATTICE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ResultView), C.c_uint32, C.c_void_p)
CONNECTION_PLUGIN_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(LatticePairs), C.c_void_p)
EXTRA_SEED_DT = np.dtype([('start', '<u2'), ('end', '<u2'), ('hash', '<i4'), ('row', '<i4', (8,))])

NODE_DT = np.dtype([('eptr', '<i4'), ('start', '<u2'), ('end', '<u2')])
UNK_DT = np.dtype([('tmpl', '<i4'), ('hash', '<i4'), ('ph0', '<u2'), ('ph1', '<u2'), ('maker', '<u2'), ('pad', '<u2')])
BEAM_DT = np.dtype([('left', '<u2'), ('beam', '<u2'), ('total', '<f4'), ('prev_node', '<u4'), ('pad', '<u4')])
GBEAM_DT = np.dtype([('left', '<u2'), ('beam', '<u2'), ('score', '<f4')])
NBEST_DT = np.dtype([('node', '<u4'), ('slot', '<u4'), ('beam', BEAM_DT), ('info', NODE_DT), ('unk', UNK_DT),
                     ('cells', '<f4', (2,))])

class Reserve(C.Structure):
    """jppgpu_reserve"""
    _fields_ = [('struct_size', C.c_uint32), ('max_sentences', C.c_uint32), ('max_total_bytes', C.c_uint64),
                ('nodes_per_byte', C.c_float), ('text_bytes_per_byte', C.c_float), ('text_host_blocks', C.c_uint32),
                ('reserved', C.c_uint32)]


class CtxStatistics(C.Structure):
    """jppgpu_ctx_statistics"""
    _fields_ = [('struct_size', C.c_uint32), ('reserved', C.c_uint32), ('one_enqueue_batches', C.c_uint64),
                ('one_enqueue_overflows', C.c_uint64), ('sized_batches', C.c_uint64), ('device_allocations', C.c_uint64)]


_libs = {}


def load_library(path=None):
    path = path or os.environ.get('JPPGPU_LIB') or DEFAULT_LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError('jppgpu: native library %s is missing -- build it with '
                           '`python -c "import __graft_entry__ as g; g.build()"`' % path)
    lib = C.CDLL(path)
    lib.jppgpu_last_error.restype = C.c_char_p
    lib.jppgpu_ctx_create.argtypes = [C.POINTER(Model), C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.jppgpu_ctx_create_shared.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.jppgpu_ctx_destroy.argtypes = [C.c_void_p]
    lib.jppgpu_analyze_batch.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.jppgpu_analyze_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                C.c_void_p, C.POINTER(C.c_void_p)]
    lib.jppgpu_result_fetch.argtypes = [C.c_void_p, C.c_int, C.POINTER(ResultView)]
    lib.jppgpu_result_fetch_nbest.argtypes = [C.c_void_p, C.c_int32, C.POINTER(NbestView)]
    lib.jppgpu_result_fetch_top1_ngrams.argtypes = [C.c_void_p, C.POINTER(NgramsView)]
    lib.jppgpu_ctx_set_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.jppgpu_analyze_batch_seeds.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, SEED_HOOK, C.c_void_p,
                                               C.POINTER(C.c_void_p)]
    lib.jppgpu_analyze_batch_scored.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.POINTER(SCORE_LATTICE_FN), C.c_void_p,
                                                C.c_uint32, C.POINTER(C.c_void_p)]
    lib.jppgpu_analyze_batch_pairs.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, CONNECTION_PLUGIN_FN, C.c_void_p,
                                               C.POINTER(C.c_void_p)]
    lib.jppgpu_result_fetch_path_ngrams.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NgramsView)]
    lib.jppgpu_result_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.jppgpu_result_release.argtypes = [C.c_void_p]
    lib.jppgpu_result_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.jppgpu_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    lib.jppgpu_ctx_reserve.argtypes = [C.c_void_p, C.POINTER(Reserve)]
    lib.jppgpu_ctx_stats.argtypes = [C.c_void_p, C.POINTER(CtxStatistics)]
    _libs[path] = lib
    return lib


class JppGpuError(RuntimeError):
    pass


def read_image(path):
    """sections of a `ref_dump export` image: list of (tag, aux, bytes)"""
    data = open(path, 'rb').read()
    if data[:8] != b'JPPGPUI1':
        raise JppGpuError('bad model image magic')
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


class Result:
    """numpy views over one fetched batch (copies owned by the native result)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.handle = handle
        self.view = None

    def _arr(self, ptr, dtype, count):
        if not ptr or count == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=count)

    def fetch(self, full=False, top1=False):
        """full: the whole lattice; top1: only the top-1 path's nodes, compacted (JPPGPU_FETCH_TOP1)"""
        v = ResultView()
        rc = self.ctx.lib.jppgpu_result_fetch(self.handle, 2 if top1 else 1 if full else 0, C.byref(v))
        if rc != 0:
            raise JppGpuError(self.ctx.lib.jppgpu_last_error().decode())
        self.view = v
        n, N, NB = v.n_sentences, v.total_nodes, v.total_boundaries
        self.n = n
        self.beam, self.gbeam = v.beam, v.global_beam
        self.status = self._arr(v.status, '<i4', n)
        self.ncp = self._arr(v.n_codepoints, '<u4', n)
        self.nnodes = self._arr(v.n_nodes, '<u4', n)
        self.node_base = self._arr(v.node_base, '<u8', n)
        self.bnd_base = self._arr(v.bnd_base, '<u8', n)
        self.path_len = self._arr(v.path_len, '<u4', n)
        self.path_nodes = self._arr(v.path_nodes, '<u4', N)
        self.nodes = self._arr(v.nodes, NODE_DT, N)
        self.unk = self._arr(v.unk, UNK_DT, N)
        if full and not top1:
            self.bnd_first = self._arr(v.bnd_first, '<u4', NB)
            self.bnd_count = self._arr(v.bnd_count, '<u4', NB)
            self.end_first = self._arr(v.end_first, '<u4', NB)
            self.end_count = self._arr(v.end_count, '<u4', NB)
            self.end_nodes = self._arr(v.end_nodes, '<u4', N)
            stride = int(v.entry_row_stride) or 8
            self.entry_rows = self._arr(v.entry_rows, '<i4', N * stride).reshape(-1, stride)
            self.patterns = self._arr(v.patterns, '<u8', N * 14).reshape(-1, 14)
            self.t0 = self._arr(v.t0_scores, '<f4', N)
            self.beams = self._arr(v.beams, BEAM_DT, N * v.beam).reshape(-1, v.beam)
            self.cells = self._arr(v.cells, '<f4', N * v.global_beam * v.num_scorers).reshape(N if v.global_beam else 0, v.global_beam, v.num_scorers)
            self.kept = self._arr(v.kept, 'u1', N)
            self.gbeam_count = self._arr(v.gbeam_count, '<u4', NB)
            self.gbeam_entries = self._arr(v.gbeam, GBEAM_DT, NB * v.global_beam).reshape(NB if v.global_beam else 0, v.global_beam)
        return self

    def fetch_nbest(self, n_best):
        """(eos slots [n, n_best], path_first [n * n_best + 1], items) of jppgpu_result_fetch_nbest"""
        v = NbestView()
        rc = self.ctx.lib.jppgpu_result_fetch_nbest(self.handle, n_best, C.byref(v))
        if rc != 0:
            raise JppGpuError(self.ctx.lib.jppgpu_last_error().decode())
        n = v.n_sentences
        first = self._arr(v.path_first, '<u8', n * n_best + 1)
        total = int(first[-1]) if n else 0
        return (self._arr(v.eos, BEAM_DT, n * n_best).reshape(n, n_best), first, self._arr(v.items, NBEST_DT, total),
                self._arr(v.n_nodes, '<u4', n))

    def fetch_top1_ngrams(self):
        """(path_first [n + 1], path_nodes [M], features [M, n_ngram]) of jppgpu_result_fetch_top1_ngrams"""
        v = NgramsView()
        rc = self.ctx.lib.jppgpu_result_fetch_top1_ngrams(self.handle, C.byref(v))
        if rc != 0:
            raise JppGpuError(self.ctx.lib.jppgpu_last_error().decode())
        n = v.n_sentences
        first = self._arr(v.path_first, '<u8', n + 1)
        total = int(first[-1]) if n else 0
        return first, self._arr(v.path_nodes, '<u4', total), self._arr(v.features, '<u4', total * v.n_ngram).reshape(total, v.n_ngram)

    def fetch_path_ngrams(self, path_first, path_nodes):
        """features [M, n_ngram] of jppgpu_result_fetch_path_ngrams for paths given in text order"""
        pf = np.ascontiguousarray(path_first, dtype=np.uint64)
        pn = np.ascontiguousarray(path_nodes, dtype=np.uint32)
        v = NgramsView()
        rc = self.ctx.lib.jppgpu_result_fetch_path_ngrams(self.handle, pf.ctypes.data, pn.ctypes.data, C.byref(v))
        if rc != 0:
            raise JppGpuError(self.ctx.lib.jppgpu_last_error().decode())
        total = int(pf[-1])
        return self._arr(v.features, '<u4', total * v.n_ngram).reshape(total, v.n_ngram)

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        rc = self.ctx.lib.jppgpu_result_stats(self.handle, C.byref(a), C.byref(b))
        if rc != 0:
            raise JppGpuError(self.ctx.lib.jppgpu_last_error().decode())
        return a.value, b.value

    def pack(self, offsets_ptr, items_ptr, cap_items):
        """packed top-1 morphemes into caller-owned device buffers (see jppgpu_result_pack)"""
        rc = self.ctx.lib.jppgpu_result_pack(self.handle, offsets_ptr, items_ptr, cap_items)
        if rc != 0:
            raise JppGpuError(self.ctx.lib.jppgpu_last_error().decode())

    def release(self):
        if self.handle:
            self.ctx.lib.jppgpu_result_release(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Context:
    """jppgpu_ctx: model resident in HBM + analysis configuration."""

    def __init__(self, image_path, beam=5, global_beam=6, right_check=1, right_beam=5,
                 max_input_bytes=4096, device=0, lib_path=None, use_rnn=None,
                 weight_perceptron=None, weight_rnn=None, rnn_nce_bias=None, dynamic_features=False, max_unk_makers=None,
                 host_scorer_weights=(), share_with=None, _no_field_storages=False):
        """use_rnn=None: run the RNN scorer iff the model image has an RNN part (what
        JumanppEnv::loadModel does); the score weights default to the model's saved
        RnnInferenceConfig (env.cc:86-100)."""
        self.lib = load_library(lib_path)
        secs = read_image(image_path)
        by = {}
        for tag, aux, payload in secs:
            by.setdefault(tag, []).append((aux, payload))
        info = struct.unpack_from('<8i', by[1][0][1], 0)
        self.num_features, self.num_data, self.num_placeholders = info[0], info[1], info[2]
        self._keep = []

        def buf(b):
            a = np.frombuffer(b, dtype=np.uint8).copy()
            self._keep.append(a)
            return a.ctypes.data, a.size

        m = Model()
        m.trie, m.trie_bytes = buf(by[2][0][1])
        m.entry_ptrs, m.entry_ptrs_bytes = buf(by[3][0][1])
        m.entry_data, m.entry_data_bytes = buf(by[4][0][1])
        if 5 not in by:
            raise JppGpuError('model image has no perceptron weights (untrained model)')
        m.weight_exponent = by[5][0][0]
        m.weights, _ = buf(by[5][0][1])
        self.weights = np.frombuffer(by[5][0][1], dtype=np.float32).copy()   # (the table as loaded; see set_weights)
        m.num_features = self.num_features
        m.num_placeholders = self.num_placeholders
        ub = by[6][0][1]
        pos = 0
        (n_unk,) = struct.unpack_from('<i', ub, pos)
        pos += 4
        makers = (UnkMaker * n_unk)()
        for i in range(n_unk):
            t, cc, pp, pr, ph, nrep = struct.unpack_from('<6i', ub, pos)
            pos += 24
            mask = 0
            for _ in range(nrep):
                (f,) = struct.unpack_from('<i', ub, pos)
                pos += 4
                mask |= 1 << f
            makers[i] = UnkMaker(t, cc, pp, pr, ph, mask)
        self._keep.append(makers)
        m.unk_makers = makers
        m.num_unk_makers = n_unk if max_unk_makers is None else min(n_unk, max_unk_makers)   # (tests: a model whose makers cannot connect every input)
        m.feature_spec, m.feature_spec_bytes = buf(by[7][0][1])
        self.has_rnn = 11 in by and any(a == 100 for a, _ in by[11])
        wp, wr = 1.0, 0.0
        if self.has_rnn:
            blk = dict(by[11])
            hdr = blk[100]
            E, order, msize, vsize, nce, unk_id, unk_c, unk_l, wp, wr, nf = struct.unpack_from('<IIQQfiffffI', hdr, 0)
            fields = struct.unpack_from('<%dI' % nf, hdr, 52)
            m.has_rnn = 1
            m.rnn_known_index, m.rnn_known_index_bytes = buf(blk[1])
            m.rnn_unk_index, m.rnn_unk_index_bytes = buf(blk[2])
            m.rnn_matrix, _ = buf(blk[3])
            m.rnn_embeddings, _ = buf(blk[4])
            m.rnn_nce_embeddings, _ = buf(blk[5])
            m.rnn_maxent, _ = buf(blk[6])
            m.rnn_layer_size, m.rnn_maxent_order, m.rnn_maxent_size, m.rnn_vocab_size = E, order, msize, vsize
            m.rnn_nce_constant = nce if rnn_nce_bias is None else rnn_nce_bias
            m.rnn_unk_id, m.rnn_unk_constant, m.rnn_unk_length = unk_id, unk_c, unk_l
            m.rnn_num_fields = nf
            for i, f in enumerate(fields):
                m.rnn_fields[i] = f
        if use_rnn is None:
            use_rnn = self.has_rnn
        if weight_perceptron is not None:
            wp = weight_perceptron
        if weight_rnn is not None:
            wr = weight_rnn
        cfg = Config(C.sizeof(Config), beam, global_beam, right_check, right_beam, max_input_bytes, device,
                     1 if use_rnn else 0, wp, wr, 1 if dynamic_features else 0, len(host_scorer_weights),
                     (C.c_float * 2)(*(list(host_scorer_weights) + [0.0, 0.0])[:2]))
        # the value storages of the feature columns (what a spec's length primitives read): SEC_FIELDS (8) names, per
        # column, its string storage (SEC_STRINGS, 9) or int-list storage (SEC_INTS, 10)
        stor = []
        if 8 in by:
            fb = by[8][0][1]
            (nfld,) = struct.unpack_from('<i', fb, 0)
            pos = 4
            strings = {a: p for a, p in by.get(9, [])}
            ints = {a: p for a, p in by.get(10, [])}
            for _ in range(nfld):
                idx, _spec, ctype, sst, ist, align, _key = struct.unpack_from('<7i', fb, pos)
                pos += 28
                for _k in range(2):
                    (ln,) = struct.unpack_from('<i', fb, pos)
                    pos += 4 + ln
                pos = (pos + 7) & ~7
                if idx < 0:
                    continue
                if ctype == 0 and sst in strings:      # FieldType::String
                    d, n = buf(strings[sst])
                    stor.append(FieldStorage(idx, 1, align, 0, d, n))
                elif ctype == 2 and ist in ints:       # FieldType::StringList
                    d, n = buf(ints[ist])
                    stor.append(FieldStorage(idx, 2, align, 0, d, n))
        if stor and not _no_field_storages:
            arr = (FieldStorage * len(stor))(*stor)
            self._keep.append(arr)
            cfg.field_storages = arr
            cfg.num_field_storages = len(stor)
        h = C.c_void_p()
        if share_with is not None:   # the other context's copy of the model in HBM (jppgpu_ctx_create_shared)
            rc = self.lib.jppgpu_ctx_create_shared(share_with.handle, C.byref(cfg), C.byref(h))
        else:
            rc = self.lib.jppgpu_ctx_create(C.byref(m), C.byref(cfg), C.byref(h))
        if rc != 0:
            raise JppGpuError('jppgpu_ctx_create failed (%d): %s' % (rc, self.lib.jppgpu_last_error().decode()))
        self.handle = h
        self.cfg = cfg

    def reserve(self, max_sentences, max_total_bytes, nodes_per_byte=0.0, text_bytes_per_byte=0.0, text_host_blocks=0):
        """jppgpu_ctx_reserve: every buffer of a batch of that size now; batches within it allocate nothing and are one enqueue"""
        r = Reserve(C.sizeof(Reserve), max_sentences, max_total_bytes, nodes_per_byte, text_bytes_per_byte, text_host_blocks, 0)
        rc = self.lib.jppgpu_ctx_reserve(self.handle, C.byref(r))
        if rc != 0:
            raise JppGpuError('jppgpu_ctx_reserve failed (%d): %s' % (rc, self.lib.jppgpu_last_error().decode()))

    def stats(self):
        """jppgpu_ctx_stats as a dict"""
        st = CtxStatistics(C.sizeof(CtxStatistics))
        rc = self.lib.jppgpu_ctx_stats(self.handle, C.byref(st))
        if rc != 0:
            raise JppGpuError('jppgpu_ctx_stats failed (%d): %s' % (rc, self.lib.jppgpu_last_error().decode()))
        return {k: int(getattr(st, k)) for k in ('one_enqueue_batches', 'one_enqueue_overflows', 'sized_batches', 'device_allocations')}

    def analyze(self, sentences):
        """sentences: list of bytes/str.  Host buffers in, Result (device resident) out."""
        enc = [s.encode('utf-8') if isinstance(s, str) else bytes(s) for s in sentences]
        offs = np.zeros(len(enc) + 1, dtype=np.uint32)
        if enc:
            offs[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64).astype(np.uint32)
        text = b''.join(enc)
        r = C.c_void_p()
        rc = self.lib.jppgpu_analyze_batch(self.handle, text, offs.ctypes.data, len(enc), C.byref(r))
        if rc != 0:
            raise JppGpuError('jppgpu_analyze_batch failed (%d): %s' % (rc, self.lib.jppgpu_last_error().decode()))
        return Result(self, r)

    def analyze_with_seeds(self, sentences, hook):
        """jppgpu_analyze_batch_seeds.  hook(seeds) -> list (one per sentence) of lists of (start, end, hash, row[<=8]);
        `seeds` is a dict of numpy arrays: status, n_codepoints, n_seeds, seed_base, seeds (NODE_DT), unk (UNK_DT)"""
        enc = [s.encode('utf-8') if isinstance(s, str) else bytes(s) for s in sentences]
        n = len(enc)
        offs = np.zeros(n + 1, dtype=np.uint32)
        if enc:
            offs[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64).astype(np.uint32)
        keep = {}

        def c_hook(_user, view_p, out_p):
            try:
                v = view_p.contents

                def arr(ptr, dt, cnt):
                    if not ptr or cnt == 0:
                        return np.zeros(0, dtype=dt)
                    return np.frombuffer((C.c_char * (np.dtype(dt).itemsize * cnt)).from_address(ptr), dtype=dt).copy()
                ns = arr(v.n_seeds, '<u4', n)
                sb = arr(v.seed_base, '<u8', n)
                total = int((sb + ns).max()) if n else 0
                view = {'status': arr(v.status, '<i4', n), 'n_codepoints': arr(v.n_codepoints, '<u4', n), 'n_seeds': ns,
                        'seed_base': sb, 'seeds': arr(v.seeds, NODE_DT, total), 'unk': arr(v.unk, UNK_DT, total)}
                extra = hook(view)
                eo = np.zeros(n + 1, dtype=np.uint32)
                flat = []
                for q in range(n):
                    flat.extend(extra[q] if extra else [])
                    eo[q + 1] = len(flat)
                es = np.zeros(len(flat), dtype=EXTRA_SEED_DT)
                for k, (st, en, h, row) in enumerate(flat):
                    es[k]['start'], es[k]['end'], es[k]['hash'] = st, en, h
                    es[k]['row'][:len(row)] = row
                keep['eo'], keep['es'] = eo, es
                out_p.contents.offsets = eo.ctypes.data
                out_p.contents.seeds = es.ctypes.data if len(flat) else None
                return 0
            except Exception as e:   # an exception must not cross the C frame
                keep['error'] = e
                return 1
        cb = SEED_HOOK(c_hook)
        r = C.c_void_p()
        rc = self.lib.jppgpu_analyze_batch_seeds(self.handle, b''.join(enc), offs.ctypes.data, n, cb, None, C.byref(r))
        if 'error' in keep:
            raise keep['error']
        if rc != 0:
            raise JppGpuError('jppgpu_analyze_batch_seeds failed (%d): %s' % (rc, self.lib.jppgpu_last_error().decode()))
        return Result(self, r)

    def _pack(self, sentences):
        enc = [s.encode('utf-8') if isinstance(s, str) else bytes(s) for s in sentences]
        offs = np.zeros(len(enc) + 1, dtype=np.uint32)
        if enc:
            offs[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64).astype(np.uint32)
        return b''.join(enc), offs, len(enc)

    def analyze_scored(self, sentences, scorers):
        """jppgpu_analyze_batch_scored: `scorers` = one callable per host scorer of the context (host_scorer_weights),
        called as f(lattice, scorer_idx, cells) with `lattice` a dict of numpy views of the full result view and `cells`
        the writable [total_nodes, global_beam, num_scorers] float32 array"""
        text, offs, n = self._pack(sentences)
        keep = []

        def wrap(f):
            def c_fn(_user, view_p, idx, cells_p):
                v = view_p.contents
                N, NB, G, S = v.total_nodes, v.total_boundaries, v.global_beam, v.num_scorers

                def arr(ptr, dt, cnt):
                    if not ptr or cnt == 0:
                        return np.zeros(0, dtype=dt)
                    return np.frombuffer((C.c_char * (np.dtype(dt).itemsize * cnt)).from_address(ptr), dtype=dt, count=cnt)
                lat = dict(n=v.n_sentences, beam=v.beam, gbeam=G, nscorers=S, status=arr(v.status, '<i4', v.n_sentences),
                           ncp=arr(v.n_codepoints, '<u4', v.n_sentences), nnodes=arr(v.n_nodes, '<u4', v.n_sentences),
                           node_base=arr(v.node_base, '<u8', v.n_sentences), bnd_base=arr(v.bnd_base, '<u8', v.n_sentences),
                           nodes=arr(v.nodes, NODE_DT, N), bnd_first=arr(v.bnd_first, '<u4', NB), bnd_count=arr(v.bnd_count, '<u4', NB),
                           end_first=arr(v.end_first, '<u4', NB), end_count=arr(v.end_count, '<u4', NB), end_nodes=arr(v.end_nodes, '<u4', N),
                           gbeam_count=arr(v.gbeam_count, '<u4', NB), gbeam_entries=arr(v.gbeam, GBEAM_DT, NB * G).reshape(-1, G),
                           beams=arr(v.beams, BEAM_DT, N * v.beam).reshape(-1, v.beam))
                cells = np.frombuffer((C.c_char * (4 * N * G * S)).from_address(cells_p), dtype='<f4', count=N * G * S).reshape(N, G, S)
                try:
                    f(lat, int(idx), cells)
                    return 0
                except Exception:   # noqa: BLE001 -- reported through the ABI's error path
                    import traceback
                    traceback.print_exc()
                    return 1
            return SCORE_LATTICE_FN(c_fn)
        fns = (SCORE_LATTICE_FN * len(scorers))(*[wrap(f) for f in scorers])
        keep.append(fns)
        r = C.c_void_p()
        rc = self.lib.jppgpu_analyze_batch_scored(self.handle, text, offs.ctypes.data, n, fns, None, len(scorers), C.byref(r))
        if rc != 0:
            raise JppGpuError('jppgpu_analyze_batch_scored failed (%d): %s' % (rc, self.lib.jppgpu_last_error().decode()))
        return Result(self, r)

    def analyze_pairs(self, sentences, plugin):
        """jppgpu_analyze_batch_pairs: plugin(lattice, penalty) fills the float32 array `penalty` [total_pairs]"""
        text, offs, n = self._pack(sentences)

        def c_fn(_user, view_p, pen_p):
            v = view_p.contents
            N, NB = v.total_nodes, v.total_boundaries

            def arr(ptr, dt, cnt):
                if not ptr or cnt == 0:
                    return np.zeros(0, dtype=dt)
                return np.frombuffer((C.c_char * (np.dtype(dt).itemsize * cnt)).from_address(ptr), dtype=dt, count=cnt)
            lat = dict(n=v.n_sentences, status=arr(v.status, '<i4', v.n_sentences), ncp=arr(v.n_codepoints, '<u4', v.n_sentences),
                       nnodes=arr(v.n_nodes, '<u4', v.n_sentences), node_base=arr(v.node_base, '<u8', v.n_sentences),
                       bnd_base=arr(v.bnd_base, '<u8', v.n_sentences), nodes=arr(v.nodes, NODE_DT, N),
                       bnd_first=arr(v.bnd_first, '<u4', NB), bnd_count=arr(v.bnd_count, '<u4', NB), end_first=arr(v.end_first, '<u4', NB),
                       end_count=arr(v.end_count, '<u4', NB), end_nodes=arr(v.end_nodes, '<u4', N), pair_base=arr(v.pair_base, '<u8', NB + 1))
            pen = arr(pen_p, '<f4', v.total_pairs)
            plugin(lat, pen)
        cb = CONNECTION_PLUGIN_FN(c_fn)
        r = C.c_void_p()
        rc = self.lib.jppgpu_analyze_batch_pairs(self.handle, text, offs.ctypes.data, n, cb, None, C.byref(r))
        if rc != 0:
            raise JppGpuError('jppgpu_analyze_batch_pairs failed (%d): %s' % (rc, self.lib.jppgpu_last_error().decode()))
        return Result(self, r)

    def analyze_device(self, d_text_ptr, d_offsets_ptr, n, total_bytes, stream=None):
        r = C.c_void_p()
        rc = self.lib.jppgpu_analyze_batch_device(self.handle, d_text_ptr, d_offsets_ptr, n, total_bytes,
                                                  stream, C.byref(r))
        if rc != 0:
            raise JppGpuError('jppgpu_analyze_batch_device failed (%d): %s'
                              % (rc, self.lib.jppgpu_last_error().decode()))
        return Result(self, r)

    def set_weights(self, weights):
        """jppgpu_ctx_set_weights: a float32 array of the model's table size"""
        w = np.ascontiguousarray(weights, dtype=np.float32)
        rc = self.lib.jppgpu_ctx_set_weights(self.handle, w.ctypes.data_as(C.c_void_p), w.size)
        if rc != 0:
            raise JppGpuError(self.lib.jppgpu_last_error().decode())

    def timings(self):
        ms = (C.c_float * 8)()
        self.lib.jppgpu_last_timings(self.handle, ms, 8)
        names = ['decode', 'seeds', 'layout', 't0', 'sweep', 'rnn', 'path', 'total']
        return dict(zip(names, list(ms)[:8]))

    def rnn_stats(self):
        """rows of the RNN hidden-state table of the last batch (rnn nodes + 2 per sentence) and ms of k_rnn_chain"""
        ms = (C.c_float * 16)()
        self.lib.jppgpu_last_timings(self.handle, ms, 16)
        return {'rows': int(ms[14]), 'chain_ms': float(ms[15])}

    def sweep_classes(self):
        """the sweep phase of the last batch by sentence class: ms and sentences of the variants staging 64 / 512 / any
        number of right nodes per boundary"""
        ms = (C.c_float * 14)()
        self.lib.jppgpu_last_timings(self.handle, ms, 14)
        return {'ms': [round(float(x), 4) for x in ms[8:11]], 'sentences': [int(x) for x in ms[11:14]]}

    def close(self):
        if self.handle:
            self.lib.jppgpu_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
