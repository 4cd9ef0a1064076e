"""att_speech.modules.beam_search — plain BeamSearch of the reference
(att_speech/modules/beam_search.py:13-182), the search reachable without an
external LM FST (`AttentionDecoderTCN.decode`, tcn.py:527-531).

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
import torch


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
