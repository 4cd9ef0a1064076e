"""Language-model FST container for the LM-fused searches
(att_speech/modules/beam_search.py: BeamSearchLM / RescoreSearchLM / GraphSearch).

The reference holds the LM as a `pywrapfst.Fst` (openfst-python 1.7.2,
modules/tcn.py:293-300) and touches only `start()`, `arcs(state)` (input-label
sorted), `final(state)` and `input_symbols()` (fst_utils.py:23-188,
tcn.py:306-327).  pywrapfst is not available here, so this is a stand-alone
container with that same surface, stored as flat numpy arrays in CSR order
(state, ilabel) so the searches can expand whole beams with array operations.

Readers: the AT&T text format (`fstprint` output: `src dst ilabel olabel
[weight]`, final states `state [weight]`) with optional symbol-table text files,
and the OpenFst binary `vector` / `standard` (tropical, float32) file format as
published in OpenFst's fst.h / vector-fst.h / symbol-table.cc.  PARITY UNPINNED
for the binary reader: no OpenFst build and no binary FST fixture exists in this
environment; it is covered by a write -> read round trip only.
"""
import struct

import numpy as np

_FST_MAGIC = 2125659606
_SYM_MAGIC = 2125658996
INF = float('inf')


class Arc(object):
    __slots__ = ('ilabel', 'olabel', 'weight', 'nextstate')

    def __init__(self, ilabel, olabel, weight, nextstate):
        self.ilabel, self.olabel, self.weight, self.nextstate = ilabel, olabel, weight, nextstate


class SymbolTable(object):
    """Iterates as (id, symbol) pairs like pywrapfst's; find() maps either way."""

    def __init__(self, pairs=()):
        self._pairs = [(int(k), str(s)) for k, s in pairs]
        self._by_sym = {s: k for k, s in self._pairs}
        self._by_id = {k: s for k, s in self._pairs}

    def __iter__(self):
        return iter(self._pairs)

    def __len__(self):
        return len(self._pairs)

    def find(self, key):
        if isinstance(key, (int, np.integer)):
            return self._by_id.get(int(key), '')
        return self._by_sym.get(key, -1)

    @classmethod
    def read_text(cls, path):
        pairs = []
        with open(path) as f:
            for line in f:
                parts = line.split()
                if len(parts) == 2:
                    pairs.append((int(parts[1]), parts[0]))
        return cls(pairs)


class LmFst(object):
    """Weighted acceptor/transducer over the tropical weight set (costs = -log p)."""

    def __init__(self, num_states, start, src, dst, ilabel, olabel, weight, final,
                 isymbols=None, osymbols=None):
        src = np.asarray(src, np.int64)
        order = np.lexsort((np.arange(len(src)), np.asarray(ilabel, np.int64), src))
        self._n = int(num_states)
        self._start = int(start)
        self.src = src[order]
        self.dst = np.asarray(dst, np.int64)[order]
        self.ilabel = np.asarray(ilabel, np.int64)[order]
        self.olabel = np.asarray(olabel, np.int64)[order]
        self.weight = np.asarray(weight, np.float64)[order]
        self.final_w = np.asarray(final, np.float64)
        assert self.final_w.shape == (self._n,)
        if len(self.src):
            assert self.src.min() >= 0 and self.src.max() < self._n
            assert self.dst.min() >= 0 and self.dst.max() < self._n
        # CSR: arcs of s are [ptr[s], ptr[s+1]); its epsilon arcs come first,
        # [ptr[s], ptr_ne[s])
        self.ptr = np.searchsorted(self.src, np.arange(self._n + 1)).astype(np.int64)
        neps = np.bincount(self.src[self.ilabel == 0], minlength=self._n).astype(np.int64)
        self.ptr_ne = self.ptr[:-1] + neps
        self._isyms, self._osyms = isymbols, osymbols
        self._eps_rank = False          # False: not computed yet; None: the epsilon graph has a cycle

    # ---- pywrapfst-like surface -------------------------------------------
    def start(self):
        return self._start

    def num_states(self):
        return self._n

    def states(self):
        return range(self._n)

    def arcs(self, state):
        for a in range(int(self.ptr[state]), int(self.ptr[state + 1])):
            yield Arc(int(self.ilabel[a]), int(self.olabel[a]), float(self.weight[a]),
                      int(self.dst[a]))

    def final(self, state):
        return float(self.final_w[state])

    def input_symbols(self):
        return self._isyms

    def output_symbols(self):
        return self._osyms

    def arcsort(self, sort_type='ilabel'):
        assert sort_type == 'ilabel'        # arcs are kept ilabel-sorted
        return self

    # ---- epsilon sub-graph -------------------------------------------------
    def eps_rank(self):
        """rank[s] = length of the longest epsilon path ending in s; every epsilon arc
        goes from a lower to a higher rank.  None if the epsilon graph has a cycle."""
        if self._eps_rank is False:
            eps = self.ilabel == 0
            es, ed = self.src[eps], self.dst[eps]
            rank = np.zeros(self._n, np.int64)
            indeg = np.bincount(ed, minlength=self._n)
            ready = list(np.nonzero((indeg == 0))[0])
            out = {}
            for s, d in zip(es.tolist(), ed.tolist()):
                out.setdefault(s, []).append(d)
            seen = 0
            while ready:
                s = int(ready.pop())
                seen += 1
                for d in out.get(s, ()):
                    rank[d] = max(rank[d], rank[s] + 1)
                    indeg[d] -= 1
                    if indeg[d] == 0:
                        ready.append(d)
            self._eps_rank = rank if seen == self._n else None
        return self._eps_rank

    # ---- text format ---------------------------------------------------------
    @classmethod
    def read_text(cls, path, isymbols=None, osymbols=None, acceptor=False):
        isy = SymbolTable.read_text(isymbols) if isinstance(isymbols, str) else isymbols
        osy = SymbolTable.read_text(osymbols) if isinstance(osymbols, str) else osymbols

        def lab(tok, table):
            try:
                return int(tok)
            except ValueError:
                k = table.find(tok)
                if k < 0:
                    raise ValueError("symbol %r not in the symbol table" % tok)
                return k
        src, dst, il, ol, w, fin = [], [], [], [], [], {}
        start, n = None, 0
        with open(path) as f:
            for line in f:
                p = line.split()
                if not p:
                    continue
                if start is None:
                    start = int(p[0])
                n = max(n, int(p[0]) + 1)
                if len(p) <= 2:
                    fin[int(p[0])] = float(p[1]) if len(p) == 2 else 0.0
                    continue
                n = max(n, int(p[1]) + 1)
                src.append(int(p[0])); dst.append(int(p[1]))
                il.append(lab(p[2], isy))
                if acceptor:
                    ol.append(il[-1]); rest = p[3:]
                else:
                    ol.append(lab(p[3], osy if osy is not None else isy)); rest = p[4:]
                w.append(float(rest[0]) if rest else 0.0)
        if start is None:
            raise ValueError("empty FST text file: %s" % path)
        final = np.full(n, INF)
        for s, v in fin.items():
            final[s] = v
        return cls(n, start, src, dst, il, ol, w, final, isy, osy if osy is not None else isy)

    # ---- OpenFst binary (vector, standard) -------------------------------------
    @staticmethod
    def _rd_str(f):
        (ln,) = struct.unpack('<i', f.read(4))
        return f.read(ln).decode('utf-8')

    @classmethod
    def _rd_symbols(cls, f):
        (magic,) = struct.unpack('<i', f.read(4))
        if magic != _SYM_MAGIC:
            raise ValueError("bad symbol table magic %d" % magic)
        cls._rd_str(f)                                   # table name
        _avail, size = struct.unpack('<qq', f.read(16))
        pairs = []
        for _ in range(size):
            s = cls._rd_str(f)
            (k,) = struct.unpack('<q', f.read(8))
            pairs.append((k, s))
        return SymbolTable(pairs)

    @classmethod
    def read(cls, path):
        with open(path, 'rb') as f:
            head = f.read(4)
            if len(head) < 4 or struct.unpack('<i', head)[0] != _FST_MAGIC:
                return cls.read_text(path)
            fsttype, arctype = cls._rd_str(f), cls._rd_str(f)
            if fsttype != 'vector' or arctype != 'standard':
                raise ValueError("unsupported FST file type %s/%s (vector/standard only)"
                                 % (fsttype, arctype))
            _version, flags = struct.unpack('<ii', f.read(8))
            _props, start, nstates, _narcs = struct.unpack('<Qqqq', f.read(32))
            isy = cls._rd_symbols(f) if flags & 1 else None
            osy = cls._rd_symbols(f) if flags & 2 else None
            src, dst, il, ol, w = [], [], [], [], []
            final = np.full(nstates, INF)
            arc_t = np.dtype([('il', '<i4'), ('ol', '<i4'), ('w', '<f4'), ('ns', '<i4')])
            for s in range(nstates):
                fw, na = struct.unpack('<fq', f.read(12))
                final[s] = fw
                a = np.frombuffer(f.read(16 * na), arc_t, na)
                src.append(np.full(na, s, np.int64)); dst.append(a['ns'].astype(np.int64))
                il.append(a['il'].astype(np.int64)); ol.append(a['ol'].astype(np.int64))
                w.append(a['w'].astype(np.float64))
            cat = (lambda xs, t: np.concatenate(xs) if xs else np.zeros(0, t))
            return cls(nstates, start, cat(src, np.int64), cat(dst, np.int64),
                       cat(il, np.int64), cat(ol, np.int64), cat(w, np.float64), final,
                       isy, osy)

    def write(self, path):
        def wr_str(f, s):
            b = s.encode('utf-8')
            f.write(struct.pack('<i', len(b))); f.write(b)

        def wr_symbols(f, table):
            f.write(struct.pack('<i', _SYM_MAGIC)); wr_str(f, 'symbols')
            keys = [k for k, _ in table]
            f.write(struct.pack('<qq', (max(keys) + 1) if keys else 0, len(keys)))
            for k, s in table:
                wr_str(f, s); f.write(struct.pack('<q', k))
        flags = (1 if self._isyms is not None else 0) | (2 if self._osyms is not None else 0)
        with open(path, 'wb') as f:
            f.write(struct.pack('<i', _FST_MAGIC))
            wr_str(f, 'vector'); wr_str(f, 'standard')
            f.write(struct.pack('<ii', 2, flags))
            f.write(struct.pack('<Qqqq', 0, self._start, self._n, len(self.src)))
            if self._isyms is not None:
                wr_symbols(f, self._isyms)
            if self._osyms is not None:
                wr_symbols(f, self._osyms)
            for s in range(self._n):
                lo, hi = int(self.ptr[s]), int(self.ptr[s + 1])
                f.write(struct.pack('<fq', self.final_w[s], hi - lo))
                for a in range(lo, hi):
                    f.write(struct.pack('<iifi', int(self.ilabel[a]), int(self.olabel[a]),
                                        float(self.weight[a]), int(self.dst[a])))
