"""att_speech.modules.beam_search — the searches of the reference
(att_speech/modules/beam_search.py): plain BeamSearch (:13-182), the search
reachable without an external LM FST (`AttentionDecoderTCN.decode`,
tcn.py:527-531), and the LM-fused BeamSearchLM (:185-363), RescoreSearchLM
(:366-403) and GraphSearch (:406-648) for an `att_speech.lm_fst.LmFst`.

Same step semantics and the same quirks, with the hypothesis re-indexing done
by device-side gathers instead of the reference's Python `batch x beam` double
loop (:108-124) and ONE host read-back per step for the finished-hypothesis
bookkeeping instead of `.item()` calls inside loops (:58-81).

Quirks kept on purpose (bit-compatible results):
  * `is_eos_best` is computed per hypothesis ([B*beam]) but indexed with the
    batch id (:73) — i.e. it looks at hypothesis `batch_id`, not at the batch's
    best beam;
  * `best_finished_scores_elements['acoustic']` aliases `best_finished_scores`
    (:30-32), so the raw EOS score overwrites the length-normalised one (:77-78).
"""
import numpy as np
import torch

from att_speech import fst_utils


class BeamSearch(object):
    def __init__(self, batch_size, beam_size, device, num_classes,
                 length_normalization, keep_eos_score=False):
        self.scores = torch.zeros(batch_size * beam_size, device=device)
        self.estimations = None
        self.finished_count = [0 for _ in range(batch_size)]
        self.best_finished = [[] for _ in range(batch_size)]
        self.best_finished_scores = [float('-inf')] * batch_size
        self.best_finished_scores_elements = {'acoustic': self.best_finished_scores}
        self.beam_size = beam_size
        self.batch_size = batch_size
        self.num_classes = num_classes
        self.length_normalization = length_normalization
        self.min_eos = None
        self.keep_eos_score = keep_eos_score
        self.coverage = None
        self.attentions = None
        self.print_debug = False
        self.gather_attentions = False

    def _save_best_finished(self, global_scores):
        """(:58-81) for each utterance: extended beam with the best EOS score."""
        B, beam = self.batch_size, self.beam_size
        scores = global_scores[:, -1].contiguous().view(B, -1)
        eos_scores = scores / (self.estimations.size(1) ** self.length_normalization)
        is_eos_best = torch.argmax(global_scores, dim=1) == global_scores.size(1) - 1
        ind = torch.argmax(eos_scores, dim=1)
        best_norm = eos_scores.gather(1, ind[:, None]).squeeze(1)
        best_raw = scores.gather(1, ind[:, None]).squeeze(1)
        host = torch.stack([is_eos_best[:B].to(best_norm.dtype), ind.to(best_norm.dtype),
                            best_norm, best_raw]).cpu()         # one read-back per step
        for b in range(B):
            if host[0, b] != 0 and self.finished_count[b] <= beam:
                self.finished_count[b] += 1
                if self.best_finished_scores[b] < float(host[2, b]):
                    # aliased lists: the raw score is what stays (:76-78)
                    self.best_finished_scores[b] = float(host[3, b])
                    self.best_finished[b] = self.estimations[b * beam + int(host[1, b])]

    def _get_topk(self, scores):
        """(:83-98)"""
        if self.beam_size < scores.size(1):
            return torch.topk(scores, self.beam_size, dim=1)
        new_scores, best_it = torch.topk(scores, scores.size(1), dim=1)
        to_repeat = self.beam_size - scores.size(1)
        no_scores = (torch.ones_like(new_scores[:, -1:]) * float('-inf')).repeat(1, to_repeat)
        new_scores = torch.cat((new_scores, no_scores), dim=1)
        best_it = torch.cat((best_it, best_it[:, -1:].repeat(1, to_repeat)), dim=1)
        return new_scores, best_it

    def step(self, logits, *args, **kwargs):
        """(:147-175) logits [1, B*beam, C] -> (new input ids [B*beam],
        state mapping [B*beam])."""
        B, beam, C = self.batch_size, self.beam_size, self.num_classes
        local_scores = torch.nn.functional.log_softmax(logits.squeeze(0), dim=1)
        global_scores = local_scores + self.scores.unsqueeze(1).repeat(1, C)
        if self.estimations is not None:
            self._save_best_finished(global_scores)
        # ignore EOS from now on (:126-133); first step: beam 0 only
        gs = global_scores[:, :-1].contiguous().view(B, -1)
        if self.estimations is None:
            gs = gs[:, :C - 1]
        new_scores, best_it = self._get_topk(gs)
        # re-index the hypotheses (:108-124), vectorised
        best_beams = best_it // (C - 1)
        best_letters = best_it % (C - 1)
        base = (torch.arange(B, device=best_it.device) * beam)[:, None]
        mapping = (base + best_beams).view(-1)
        if self.estimations is None:
            est = torch.zeros((B * beam, 0), dtype=torch.long, device=best_it.device)
        else:
            est = self.estimations[mapping]
        self.estimations = torch.cat((est, best_letters.reshape(-1, 1)), dim=1)
        self.scores = new_scores.reshape(-1)
        return best_letters.reshape(-1), mapping

    def has_finished(self):
        return all(self.finished_count[i] >= self.beam_size for i in range(self.batch_size))

    def get_graph(self):
        return None

    # ---- pieces shared with the LM-fused searches (:100-145) -------------------
    def to_text(self, est):
        itos = getattr(self, 'itos', None)
        if itos is None:
            return ' '.join(str(int(e)) for e in est)
        return ''.join(itos[e] if itos[e] != '<spc>' else ' ' for e in est)

    def _do_ignore_eos(self, global_scores):
        global_scores = global_scores[:, :-1].contiguous().view(self.batch_size, -1)
        if self.estimations is None:
            global_scores = global_scores[:, :self.num_classes - 1]
        return global_scores

    def _compute_new_beam(self, best_it):
        """(:108-124) with the `batch x beam` loop as two gathers; appends the new
        letters to the re-indexed hypotheses."""
        B, beam, C = self.batch_size, self.beam_size, self.num_classes
        best_beams = best_it // (C - 1)
        best_letters = best_it % (C - 1)
        base = (torch.arange(B, device=best_it.device) * beam)[:, None]
        mapping = (base + best_beams).view(-1)
        if self.estimations is None:
            est = torch.zeros((B * beam, 0), dtype=torch.long, device=best_it.device)
        else:
            est = self.estimations[mapping]
        return mapping, torch.cat((est, best_letters.reshape(-1, 1)), dim=1), best_letters

    def _get_eos_score_from_previous_frame(self, unnormalized_local_scores):
        if self.min_eos is not None:
            unnormalized_local_scores[:, -1] = torch.where(
                unnormalized_local_scores[:, -1] > self.min_eos,
                unnormalized_local_scores[:, -1], self.min_eos)
        self.min_eos = unnormalized_local_scores[:, -1]
        return unnormalized_local_scores

    def _update_eos_scores_with_new_beam(self, beam_mapping):
        self.min_eos = self.min_eos[beam_mapping]


class BeamSearchLM(BeamSearch):
    """Beam search with shallow LM fusion and a coverage term (:185-363); one
    utterance at a time.  A hypothesis carries a bag {LM state: cost}; one step
    pushes the bags of ALL beams through ALL labels in one batched array expansion
    (fst_utils.expand_all_batched) instead of a Python loop over beams x arcs, and
    only the bags of the hypotheses that survive the top-k are turned into dicts."""

    def __init__(self, lm, lm_weight, alphabet_mapping, min_attention_pos,
                 coverage_tau, coverage_weight, *args, **kwargs):
        super(BeamSearchLM, self).__init__(*args, **kwargs)
        self.lm = lm
        self.fst_states = [{self.lm.start(): 0} for _ in range(self.beam_size)]
        self.alphabet_mapping = alphabet_mapping
        self.lm_weight = lm_weight
        self.finished = []
        self.min_attention_pos = min_attention_pos
        self.coverage_tau = coverage_tau
        self.coverage_weight = coverage_weight
        self.best_finished_scores = [float('-inf')] * self.batch_size
        self.best_finished_scores_elements = {'acoustic': [0], 'lm': [0]}
        if self.coverage_weight > 0:
            self.best_finished_scores_elements['coverage'] = [0]
        assert self.batch_size == 1

    class _Bags(object):
        """bags of one step, sorted by bag id = beam * num_classes + LM label"""

        def __init__(self, bag, st, w, num_classes, mapping):
            self.bag, self.st, self.w = bag, st, w
            self.num_classes, self.mapping = num_classes, mapping

        def get(self, beam, letter):
            b = beam * self.num_classes + self.mapping[letter]
            lo, hi = np.searchsorted(self.bag, [b, b + 1])
            return dict(zip(self.st[lo:hi].tolist(), self.w[lo:hi].tolist()))

    def _step_lm(self):
        """(:209-226) -lm_weight * cost of every (beam, letter) extension."""
        C = self.num_classes
        lm_scores = torch.zeros((self.beam_size, C))
        if self.lm_weight == 0:
            return lm_scores, self.fst_states
        sizes = [len(d) for d in self.fst_states]
        grp = np.repeat(np.arange(self.beam_size), sizes)
        st = np.fromiter((k for d in self.fst_states for k in d), np.int64, sum(sizes))
        w = np.fromiter((v for d in self.fst_states for v in d.values()), np.float64, sum(sizes))
        bag, st, w = fst_utils.expand_all_batched(self.lm, C, grp, st, w, True)
        cost = np.full(self.beam_size * C, np.inf)
        if bag.size:
            ub, red = fst_utils._reduce_by_key(bag, w, True)
            cost[ub] = red
        mapping = np.asarray(self.alphabet_mapping, np.int64)
        nxt = np.minimum(1e20, cost.reshape(self.beam_size, C)[:, mapping])
        lm_scores = torch.from_numpy((-self.lm_weight * nxt).astype(np.float32))
        return lm_scores, self._Bags(bag, st, w, C, mapping)

    def _finish_candidates(self, total_scores, att_weights):
        """per-beam (normalised EOS score, may finish) with one host read-back (:236-246)"""
        min_pos = self.min_attention_pos * att_weights.size(0)
        eos = total_scores[:, -1] / (self.estimations.size(1) ** self.length_normalization)
        far = att_weights.argmax(dim=0) > min_pos
        eos_best = torch.argmax(total_scores, dim=1) == total_scores.size(1) - 1
        host = torch.stack([eos, (far & eos_best).to(eos.dtype)]).cpu()
        return host[0], host[1] != 0

    def _set_best(self, score_elements):
        if self.finished[0][0] > self.best_finished_scores[0]:
            self.best_finished_scores[0] = self.finished[0][0]
            self.best_finished[0] = self.finished[0][1]
            self.best_finished_scores_elements = {
                k: [v[self.finished[0][2], -1].item()] for k, v in score_elements.items()}

    def _add_finished(self, total_scores, score_elements, att_weights):
        """(:228-268) hypotheses whose best continuation is EOS, that look far enough
        into the utterance, join the finished list (kept sorted, beam_size long)."""
        eos, ok = self._finish_candidates(total_scores, att_weights)
        finish_mask = [False] * eos.size(0)
        added = False
        for beam in range(eos.size(0)):
            if ok[beam] and eos[beam].item() > -1e10:
                finish_mask[beam] = True
                self.finished += [(eos[beam], self.estimations[beam], beam)]
                added = True
                if self.print_debug:
                    print('Added to finshed {} {}'.format(
                        self.to_text(self.estimations[beam]), eos[beam]))
        if self.finished and added:
            self.finished = sorted(self.finished, key=lambda x: x[0].item(),
                                   reverse=True)[:self.beam_size]
            self._set_best(score_elements)
        return finish_mask

    def _score(self, logits, att_weights):
        """acoustic + LM + coverage scores of every (beam, letter) (:270-310)"""
        if self.coverage_weight > 0:
            if self.coverage is None:
                self.coverage = att_weights.clone()
            else:
                self.coverage += att_weights
        local_scores = logits.squeeze(0)
        if self.keep_eos_score:
            local_scores = self._get_eos_score_from_previous_frame(local_scores)
        local_scores = torch.nn.functional.log_softmax(local_scores, dim=1)
        acoustic_scores = local_scores + self.scores.unsqueeze(1).repeat(1, self.num_classes)
        lm_scores, all_fst_states = self._step_lm()
        lm_scores = lm_scores.to(acoustic_scores.device)
        score_elements = {'acoustic': acoustic_scores.clone(), 'lm': lm_scores.clone()}
        total_scores = acoustic_scores + lm_scores
        if self.coverage_weight > 0:
            coverages = (self.coverage > self.coverage_tau).sum(dim=0).float()
            coverage_scores = self.coverage_weight * coverages.unsqueeze(1).repeat(
                1, self.num_classes)
            total_scores += coverage_scores
            score_elements['coverage'] = coverage_scores
        return acoustic_scores, total_scores, score_elements, all_fst_states

    def _select(self, flat_scores, best_it):
        """(:318-324) scores of the chosen extensions; padding slots are -inf"""
        new_scores = flat_scores[:, best_it[0]]
        if self.beam_size >= flat_scores.size(1):
            new_scores[:, -(self.beam_size - flat_scores.size(1)):] = float('-inf')
        return new_scores.view(-1)

    def _new_fst_states(self, all_fst_states, best_it):
        if self.lm_weight == 0:
            return []
        C = self.num_classes
        return [all_fst_states.get(ind // (C - 1), ind % (C - 1)) for ind in best_it[0].tolist()]

    def _reindex(self, new_beam_mapping):
        if self.keep_eos_score:
            self._update_eos_scores_with_new_beam(new_beam_mapping)
        if self.coverage_weight > 0:
            self.coverage = self.coverage[:, new_beam_mapping]
        if self.attentions is not None:
            self.attentions = self.attentions[:, new_beam_mapping, :]

    def step(self, logits, att_weights, print_lm=None):
        """(:270-356)"""
        if self.gather_attentions:
            if self.attentions is None:
                self.attentions = att_weights.clone().unsqueeze(-1)
            else:
                self.attentions = torch.cat((self.attentions, att_weights.unsqueeze(-1)), dim=-1)
        acoustic_scores, total_scores, score_elements, all_fst_states = self._score(
            logits, att_weights)
        if self.estimations is not None:
            self._add_finished(total_scores, score_elements, att_weights)
        total_scores = self._do_ignore_eos(total_scores)        # ignore EOS from now on
        _, best_it = self._get_topk(total_scores)
        acoustic_scores = acoustic_scores[:, :-1].contiguous().view(self.batch_size, -1)
        self.scores = self._select(acoustic_scores, best_it)
        self.fst_states = self._new_fst_states(all_fst_states, best_it)
        new_beam_mapping, self.estimations, best_letters = self._compute_new_beam(best_it)
        self._reindex(new_beam_mapping)
        if self.print_debug:
            print("%s a:%.3f l:%.f c:%.3f (%d)" % (
                self.to_text(self.estimations[0]), self.scores[0],
                -fst_utils.reduce_weights(self.fst_states[0].values(), True)
                if self.lm_weight > 0 else 0,
                (self.coverage > self.coverage_tau).sum(0)[0].item()
                if self.coverage is not None else 0, len(self.finished)))
        return best_letters.view(-1), new_beam_mapping

    def debug_estimations(self):
        for est in self.estimations:
            print(self.to_text(est))

    def has_finished(self):
        return len(self.finished) >= self.beam_size


class RescoreSearchLM(BeamSearchLM):
    """Forced decoding of a given sentence with the fused score (:366-403)."""

    def __init__(self, sentence, *args, **kwargs):
        super(RescoreSearchLM, self).__init__(*args, **kwargs)
        self.sentence = sentence
        self.gather_attentions = True
        assert self.beam_size == 1

    def _get_topk(self, scores):
        let_id = self.estimations.size(1) if self.estimations is not None else 0
        cur_id = self.sentence[let_id] if let_id < len(self.sentence) else 0
        return (scores[:, cur_id:(cur_id + 1)],
                torch.LongTensor([[cur_id]]).to(scores.device))

    def _add_finished(self, global_scores, score_elements, att_weights):
        if self.estimations.size(1) == len(self.sentence):
            eos = (global_scores[:, -1] /
                   (self.estimations.size(1) ** self.length_normalization)).cpu()
            self.finished += [(eos[0], self.estimations[0], 0)]
            self._set_best(score_elements)


class GraphSearch(BeamSearchLM):
    """BeamSearchLM that merges hypotheses whose recent history (hash_dec), LM
    state set and attention agree, keeping a graph of the merges (:406-648)."""

    def __init__(self, hash_dec, merge_threshold, *args, **kwargs):
        super(GraphSearch, self).__init__(*args, **kwargs)
        self.graph = [{} for _ in range(self.batch_size)]
        self.hash_dec = hash_dec
        self.merge_threshold = merge_threshold

    def att_prod(self, x, y):
        return torch.sum(torch.min(x, y))

    def is_prefix(self, l1, l2):
        if len(l1) > len(l2):
            return False
        return bool((l1 == l2[:len(l1)]).all())

    def step(self, logits, att_weights, print_lm=None):
        """(:424-596)"""
        beam = self.beam_size
        acoustic_scores, total_scores, score_elements, all_fst_states = self._score(
            logits, att_weights)
        if self.estimations is not None:
            finish_mask = self._add_finished(total_scores, score_elements, att_weights)
        else:
            finish_mask = [False] * beam
        if beam > 1 and self.estimations is not None:
            est_host = self.estimations.cpu()
            for beam_id in range(beam):
                if not finish_mask[beam_id]:
                    continue
                li = self.graph[0].get(self.hash_dec(est_host[beam_id]), [])
                for i, (score, atts, (fst, fin, cov), ests, uplink) in enumerate(li):
                    if ests.shape == est_host[beam_id].shape and bool((ests == est_host[beam_id]).all()):
                        li[i] = (score, atts, (fst, True, cov), ests, uplink)

        total_scores = self._do_ignore_eos(total_scores)        # ignore EOS from now on
        _, best_it = self._get_topk(total_scores)
        acoustic_scores = acoustic_scores[:, :-1].contiguous().view(self.batch_size, -1)
        new_scores = self._select(acoustic_scores, best_it)
        new_tot_scores = self._select(total_scores, best_it)
        self.fst_states = new_fst_states = self._new_fst_states(all_fst_states, best_it)
        new_beam_mapping, new_estimations, best_letters = self._compute_new_beam(best_it)
        self.estimations = new_estimations
        self._reindex(new_beam_mapping)

        if beam > 1:
            # the merge bookkeeping runs on host copies (one transfer per step)
            ns, nt = new_scores.cpu(), new_tot_scores.cpu()
            est_host, att_host = new_estimations.cpu(), att_weights.detach().cpu()
            norm = new_estimations.size(1) ** self.length_normalization
            for cur in range(beam):
                if ns[cur] == float('-inf'):
                    continue
                hist_hash = self.hash_dec(est_host[cur])
                li = self.graph[0].get(hist_hash, [])
                new_uplink = None
                for i, (score, atts, (fst, fin, cov), ests, uplink) in enumerate(li):
                    if uplink is not None:
                        continue                                 # dead branch
                    if new_fst_states and set(new_fst_states[cur].keys()) != fst:
                        continue                                 # different LM state
                    if self.att_prod(atts, att_host[:, cur]) < self.merge_threshold:
                        continue                                 # a different branch
                    if score / len(ests) ** self.length_normalization >= nt[cur] / norm:
                        ns[cur] = float('-inf')                  # the old branch is better
                        nt[cur] = float('-inf')
                        new_uplink = i
                        break
                    li[i] = (score, atts, (fst, fin, cov), ests, len(li))
                    for oth in range(beam):                      # drop its descendants
                        if oth != cur and self.is_prefix(ests, est_host[oth]):
                            ns[oth] = float('-inf')
                            nt[oth] = float('-inf')
                # (a VIEW of nt, like the reference's: a branch dropped later in this step reads -inf)
                li.append((nt[cur], att_host[:, cur],
                           (set(new_fst_states[cur].keys()) if new_fst_states else set(),
                            False, None),
                           est_host[cur], new_uplink))
                self.graph[0][hist_hash] = li
            new_scores = ns.to(new_scores.device)
        self.scores = new_scores
        return best_letters.view(-1), new_beam_mapping

    def get_graph(self):
        """(:598-648) vertices (hash, letter, score, coverage, finished) and edges
        (parent hash, hash, 'normal' | 'merged') per utterance."""
        for hmap in self.graph:
            for _, li in hmap.items():
                for i in range(len(li)):
                    if li[i][4] is not None:                     # follow uplinks to the sink
                        t = i
                        while li[t][4] is not None:
                            t = li[t][4]
                        li[i] = li[i][:4] + (t,)
                for i in range(len(li)):
                    li[i] = li[i][:5] + (hash(tuple(li[i][3].tolist())), li[i][3][-1])
        G = []
        for hmap in self.graph:
            V = [(hash(()), '<sos>', 0.0, 0., False)]
            valid = {hash(())}
            E = []
            for _, li in hmap.items():
                for sc, atts, (fsts, fin, cov), ests, uplink, ests_hash, label in li:
                    if uplink is None:
                        valid.add(ests_hash)
                        V.append((ests_hash, label.item(), sc.item(), cov, fin))
            for _, li in hmap.items():
                for sc, atts, (fsts, fin, cov), ests, uplink, ests_hash, label in li:
                    parent = hash(tuple(ests[:-1].tolist()))
                    me, kind = ests_hash, 'normal'
                    if uplink is not None:
                        me, kind = li[uplink][5], 'merged'
                    if parent in valid and me in valid:
                        E.append((parent, me, kind))
            G.append({'V': V, 'E': E})
        return G


class DeviceBeamSearch(object):
    """Plain BeamSearch (reference beam_search.py:13-182) with ALL of its state on the MI355X
    and no host read-back inside a step (`asr_beam_step_f32`): running scores, label
    histories (double-buffered `[B*beam, Lcap]`), per-utterance finished counts and best
    finished hypotheses.  `step` only enqueues a launch; `poll_finished` reads the
    device-side flag (the caller decides how often); `finalize` copies the results into the
    attributes the reference's object exposes (`finished_count`, `best_finished`,
    `best_finished_scores`, `best_finished_scores_elements`, `estimations`, `scores`)."""

    def __init__(self, batch_size, beam_size, device, num_classes, length_normalization,
                 max_steps):
        from att_speech import _native
        self._native = _native
        self.batch_size, self.beam_size, self.num_classes = batch_size, beam_size, num_classes
        self.length_normalization = length_normalization
        hyps, cap = batch_size * beam_size, max_steps + 1
        i32 = dict(dtype=torch.int32, device=device)
        self._scores = [torch.zeros(hyps, device=device), torch.zeros(hyps, device=device)]
        self._est = [torch.zeros((hyps, cap), **i32), torch.zeros((hyps, cap), **i32)]
        self._state = {
            'finished_count': torch.zeros(batch_size, **i32),
            'best_score': torch.full((batch_size,), float('-inf'), device=device),
            'best_len': torch.zeros(batch_size, **i32),
            'best_tokens': torch.zeros((batch_size, cap), **i32),
            'new_input': torch.zeros(hyps, **i32), 'parent': torch.zeros(hyps, **i32),
            'done': torch.zeros(3, **i32)}
        self._step = 0
        self.coverage = None
        self.print_debug = False
        self.estimations = None
        self.scores = self._scores[0]

    def step(self, logits, *args, **kwargs):
        """logits [1, B*beam, C] or [B*beam, C] -> (chosen labels, parent hypothesis) as
        int32 device tensors (valid until the next step)."""
        s = self._step
        logits = logits.reshape(-1, self.num_classes).contiguous()
        len_div = float(s ** self.length_normalization) if s > 0 else 1.0
        self._native.beam_step(logits, self._scores[s & 1], self._scores[(s + 1) & 1],
                               self._est[s & 1], self._est[(s + 1) & 1], s, self.batch_size,
                               self.beam_size, len_div, self._state)
        self._step = s + 1
        return self._state['new_input'], self._state['parent']

    def poll_finished(self):
        return bool(int(self._state['done'][0].item()))

    has_finished = poll_finished

    def get_graph(self):
        return None

    def finalize(self):
        st = self._state
        done = st['done'].cpu().tolist()
        eff = int(done[2])                       # steps that took effect
        self.finished_count = st['finished_count'].cpu().tolist()
        lens = st['best_len'].cpu().tolist()
        toks = st['best_tokens'].cpu().long()
        self.best_finished = [toks[b, :lens[b]] if lens[b] > 0 else [] for b in range(self.batch_size)]
        self.best_finished_scores = [float(v) for v in st['best_score'].cpu().tolist()]
        self.best_finished_scores_elements = {'acoustic': self.best_finished_scores}
        self.estimations = self._est[eff & 1][:, :eff].long()
        self.scores = self._scores[eff & 1]
        return self
