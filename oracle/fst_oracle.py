"""oracle/fst_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (pure Python + numpy) of the reference's graph construction for
the CTC training / decoding lattices.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

The reference builds these graphs with pywrapfst (openfst-python 1.7.2,
environment.yml:91), which is absent from /root/reference and from this image.
The few OpenFst operations the path needs (add_state/add_arc, arcsort,
compose of an output-epsilon transducer with an epsilon-free acceptor,
input-determinism check, arc iteration) are restated here from their published
semantics as a tiny `MiniFst`; everything above them follows the reference
line by line (citations relative to /root/reference/att_speech/).

Parity status: UNPINNED at the OpenFst boundary — the reference holds no golden
graph matrices and pywrapfst cannot run here.  State numbering produced by
OpenFst's compose is therefore not reproduced bit-for-bit; all lattice results
are invariant to it except for fp32 summation order inside logsumexp.  The
graphs are validated through loss identities instead
(tests/test_graph_identities.py): mono lattices against torch F.ctc_loss and
the reference's dense get_CTC_matrices_mono + RawGenericCTCBatch
(ctc_losses.py:67-94,327-390, golden fixtures), bigram lattices against the
reference's dense get_CTC_matrices_bicontext fixtures.
"""
from collections import deque

import numpy as np


class MiniFst(object):
    """Just enough of pywrapfst.Fst: states, arcs (ilabel, olabel, w, next),
    finals, start.  Weights live in the log semiring as plain floats
    (Weight.One == 0.0, Weight.Zero == +inf)."""

    def __init__(self):
        self.arcs = []      # per state: list of (ilabel, olabel, weight, nextstate)
        self.final = []     # per state: final weight (inf = not final)
        self.start = -1

    def add_state(self):
        self.arcs.append([])
        self.final.append(float('inf'))
        return len(self.arcs) - 1

    def set_start(self, s):
        self.start = s

    def set_final(self, s, w=0.0):
        self.final[s] = w

    def add_arc(self, s, ilabel, olabel, weight, nextstate):
        self.arcs[s].append((ilabel, olabel, weight, nextstate))

    def num_states(self):
        return len(self.arcs)

    def arcsort(self, sort_type):
        # OpenFst ILabelCompare / OLabelCompare: key (label, other label)
        if sort_type == 'ilabel':
            key = lambda a: (a[0], a[1])
        else:
            key = lambda a: (a[1], a[0])
        for s in range(len(self.arcs)):
            self.arcs[s].sort(key=key)
        return self

    def is_input_deterministic(self):
        for arcs in self.arcs:
            labs = [a[0] for a in arcs]
            if len(set(labs)) != len(labs):
                return False
        return True


def build_chain_fst(labels):
    """fst_utils.py:195-219 — acceptor of the label string."""
    C = MiniFst()
    s = C.add_state()
    C.set_start(s)
    for l in labels:
        l = int(l)
        s_next = C.add_state()
        C.add_arc(s, l, l, 0.0, s_next)
        s = s_next
    C.set_final(s)
    C.arcsort('ilabel')
    return C


def build_ctc_mono_decoding_fst(S):
    """fst_utils.py:679-726."""
    CTC = MiniFst()
    for s in range(S):
        s1 = CTC.add_state()
        assert s == s1
        CTC.set_final(s1)
    CTC.set_start(0)
    for s in range(S):
        CTC.add_arc(s, s + 1, 0, 0.0, s)             # self-loop, no output (:706)
        for s_next in range(S):
            if s_next == s:
                continue
            CTC.add_arc(s, s_next + 1, s_next, 0.0, s_next)   # (:711)
    CTC.arcsort('olabel')
    return CTC


def build_ctc_bigram_decoding_fst(S, allow_nonblank_selfloops=True,
                                  use_contextual_blanks=False,
                                  loop_using_symbol_repetitions=False,
                                  eval_repeats_in_context=False):
    """fst_utils.py:729-835."""
    assert not (eval_repeats_in_context and loop_using_symbol_repetitions)
    assert not eval_repeats_in_context
    if loop_using_symbol_repetitions:
        allow_nonblank_selfloops = False
        use_contextual_blanks = True

    CTC = MiniFst()
    for s in range(S ** 2):
        s1 = CTC.add_state()
        assert s == s1
        CTC.set_final(s1)
    CTC.set_start(0)

    def get_input_sym(c, let):
        if let != 0 or use_contextual_blanks:
            return c * S + let + 1                       # (:778)
        return 0 + 1                                     # global blank (:781)

    for s1 in range(S ** 2):
        c1 = s1 // S
        l1 = s1 % S
        self_loop = None
        if allow_nonblank_selfloops and l1 != 0 and c1 != l1:
            CTC.add_arc(s1, get_input_sym(c1, l1), 0, 0.0, s1)   # (:791-795)
            self_loop = s1
        c2 = c1 if l1 == 0 else l1                       # (:798-801)
        for l2 in range(S):
            s2 = c2 * S + l2
            assert not self_loop == s2
            if (l2 == 0 or s1 == s2 or
                    (loop_using_symbol_repetitions and l1 == l2)):
                out_s = 0
            else:
                out_s = l2
            CTC.add_arc(s1, get_input_sym(c2, l2), out_s, 0.0, s2)  # (:814-816)
    CTC.arcsort('olabel')
    return CTC


def compose(a, b):
    """OpenFst Compose(a, b) for the only case the path uses
    (fst_utils.py:604): `a` a transducer whose output side may carry epsilons
    (olabel 0), `b` an epsilon-free acceptor.  An output-epsilon arc of `a`
    advances `a` alone; any other arc must match an arc of `b` on
    a.olabel == b.ilabel.  States are numbered in discovery order, visiting
    states by increasing id and arcs in `a`'s stored order (what copying a lazy
    ComposeFst into a VectorFst does); the result is trimmed to coaccessible
    states like pywrapfst's compose(connect=True)."""
    out = MiniFst()
    ids = {}
    queue = deque()

    def get_id(pair):
        if pair not in ids:
            ids[pair] = out.add_state()
            queue.append(pair)
        return ids[pair]

    out.set_start(get_id((a.start, b.start)))
    while queue:
        sa, sb = pair = queue.popleft()
        sid = ids[pair]
        fa, fb = a.final[sa], b.final[sb]
        if np.isfinite(fa) and np.isfinite(fb):
            out.set_final(sid, fa + fb)
        for (il, ol, w, na) in a.arcs[sa]:
            if ol == 0:
                out.add_arc(sid, il, 0, w, get_id((na, sb)))
            else:
                for (il2, ol2, w2, nb) in b.arcs[sb]:
                    if il2 == ol:
                        out.add_arc(sid, il, ol2, w + w2, get_id((na, nb)))
    return _connect(out)


def _connect(g):
    n = g.num_states()
    rev = [[] for _ in range(n)]
    for s in range(n):
        for a in g.arcs[s]:
            rev[a[3]].append(s)
    coacc = [np.isfinite(f) for f in g.final]
    stack = [s for s in range(n) if coacc[s]]
    while stack:
        s = stack.pop()
        for p in rev[s]:
            if not coacc[p]:
                coacc[p] = True
                stack.append(p)
    if all(coacc):
        return g
    remap = {}
    out = MiniFst()
    for s in range(n):
        if coacc[s]:
            remap[s] = out.add_state()
            out.final[remap[s]] = g.final[s]
    for s in range(n):
        if coacc[s]:
            for (il, ol, w, ns) in g.arcs[s]:
                if coacc[ns]:
                    out.add_arc(remap[s], il, ol, w, remap[ns])
    out.set_start(remap[g.start])
    return out


def fst_to_matrices(g, out_edges=True, nc_weight=-1e20):
    """fst_utils.py:222-294 — adjacency lists as padded matrices.
    Returns (states [N,K] i64, ilabels [N,K] i64, weights [N,K] f32,
    terminal [N,1] f32)."""
    if g.start != 0:
        raise ValueError("FST starting state is not 0, but %d" % (g.start,))
    if not g.is_input_deterministic():
        raise ValueError("FST is not deterministic")
    nc_weight = float(nc_weight)
    n = g.num_states()
    edges = [[] for _ in range(n)]
    terminal_mat = np.full((n, 1), nc_weight, dtype=np.float32)
    for prevstate in range(n):
        term_weight = -float(g.final[prevstate])                     # (:263)
        terminal_mat[prevstate, 0] = (term_weight if np.isfinite(term_weight)
                                      else nc_weight)
        for (il, ol, w, nextstate) in g.arcs[prevstate]:
            ilabel = il - 1                                          # (:270)
            weight = -float(w)                                       # (:271)
            if ilabel < 0:
                raise ValueError(
                    "FST has eps-transitions (state=%d)" % (prevstate,))
            if out_edges:
                edges[prevstate].append((nextstate, ilabel, weight))
            else:
                edges[nextstate].append((prevstate, ilabel, weight))
    k = max(len(e) for e in edges)
    states_mat = np.zeros((n, k), dtype=np.int64)
    ilabels_mat = np.zeros((n, k), dtype=np.int64)
    weights_mat = np.full((n, k), nc_weight, dtype=np.float32)
    for s1, arcs in enumerate(edges):
        for i, (s2, ilabel, weight) in enumerate(sorted(arcs)):       # (:285)
            states_mat[s1, i] = s2
            ilabels_mat[s1, i] = ilabel
            weights_mat[s1, i] = weight
    return states_mat, ilabels_mat, weights_mat, terminal_mat


def batch_training_graph_matrices(matrices, nc_weight=-1e20):
    """fst_utils.py:491-521."""
    bs = len(matrices)
    max_n = max(m[0].shape[0] for m in matrices)
    max_ks = [max(m[i].shape[1] for m in matrices)
              for i in range(len(matrices[0]))]
    batched = []
    for i, m in enumerate(matrices[0]):
        batched.append(np.full(
            (bs, max_n, max_ks[i]),
            0 if m.dtype == np.int64 else nc_weight, dtype=m.dtype))
    for b, ms in enumerate(matrices):
        for i, m in enumerate(ms):
            batched[i][b, :m.shape[0], :m.shape[1]] = m
    return batched


class CTCGraphGen(object):
    """fst_utils.py:546-676,1053-1068 restricted to context_order 1 and 2
    without a grammar FST."""

    def __init__(self, num_symbols, context_order=1, nc_weight=-1e20,
                 for_forward_only=False, graph_build_args=None):
        assert context_order in (1, 2)
        self.num_symbols = num_symbols
        self.context_order = context_order
        self.num_classes = num_symbols ** context_order
        self.nc_weight = nc_weight
        self.for_forward_only = for_forward_only
        self.graph_build_args = graph_build_args or {}
        if context_order == 1:                                       # (:1061-1065)
            self.decoding_fst = build_ctc_mono_decoding_fst(num_symbols)
        else:
            self.decoding_fst = build_ctc_bigram_decoding_fst(
                num_symbols, **self.graph_build_args)

    def get_transcript_fst(self, labels):                            # (:592-601)
        labels = np.asarray(labels)
        if np.any(labels > self.num_symbols):
            labels = labels % self.num_symbols
        return build_chain_fst(labels)

    def get_training_fst(self, labels):                              # (:603-605)
        return compose(self.decoding_fst, self.get_transcript_fst(labels))

    def get_training_matrices(self, labels):                         # (:647-660)
        train_fst = self.get_training_fst(labels)
        matrices = fst_to_matrices(train_fst, out_edges=False,
                                   nc_weight=self.nc_weight)
        if not self.for_forward_only:
            matrices += fst_to_matrices(train_fst, out_edges=True,
                                        nc_weight=self.nc_weight)
        return matrices

    def get_training_matrices_batch(self, labels, label_lens):       # (:607-613)
        matrices = [self.get_training_matrices(labels[i][:label_lens[i]])
                    for i in range(len(labels))]
        return batch_training_graph_matrices(matrices, self.nc_weight)

    def get_decoding_matrices(self):                                 # (:662-676)
        mats = [m[None] for m in fst_to_matrices(
            self.decoding_fst, out_edges=False, nc_weight=self.nc_weight)]
        if not self.for_forward_only:
            mats += [m[None] for m in fst_to_matrices(
                self.decoding_fst, out_edges=True, nc_weight=self.nc_weight)]
        return mats


def read_out_olabels(dec_fst, ilabels):
    """advanced_decoder.py:556-571: compose(chain(ilabels+1), dec_fst),
    shortestpath, collect non-epsilon olabels.  dec_fst is input-deterministic,
    so the composition is the single path that follows the ilabels from the
    start state."""
    out = []
    s = dec_fst.start
    for il in ilabels:
        il = int(il) + 1
        for (ail, aol, w, ns) in dec_fst.arcs[s]:
            if ail == il:
                if aol > 0:
                    out.append(aol)
                s = ns
                break
        else:
            return []   # not accepted: empty composition, no arcs to walk
    return out


def process_sequence(frames, frames_len, blanks, num_symbols):
    """CTCDecoderAdvanced.process_sequence, default (fix_greedy_decoder=False)
    branch, bug-compatible (advanced_decoder.py:385-391): the `or` makes the
    adjacent-frame test vacuous for i != 0, and for i == 0 compares with the
    LAST element of the (padded) frame row."""
    ret = []
    seq = frames[:frames_len]
    for i, char in enumerate(seq):
        char = int(char)
        if char not in blanks and (i != 0 or char != int(frames[i - 1])):
            if not ret or (ret[-1] % num_symbols != char % num_symbols):
                ret.append(char)
    return ret
