"""att_speech.modules.tcn — the stage-2 decoder of the reference
(att_speech/modules/tcn.py): a causal dilated-convolution language model over
the label history (`TCN` of `TemporalBlock`s, :46-116), a location-aware
`LocalAttention` (:119-230) and `AttentionDecoderTCN` (:233-585) with the
teacher-forced training `forward` (:357-440) and the step-wise beam `decode`
(:442-585; plain BeamSearch, or with `lm_file` the LM-fused BeamSearchLM /
RescoreSearchLM / GraphSearch over an att_speech.lm_fst.LmFst).

Kept from the reference: class names, constructor keywords, the returned dicts
and the checkpoint keys (`tcn.network.<i>.net.conv<j>.{bias,weight_g,weight_v}`,
`attn.{encoded_to_hidden,hidden_to_score,lm_to_kernel,lm_to_global,
encoded_to_init_weights}.*`, `combined_to_output.{0,3}.*`, ...).  Organisation and
arithmetic layout are this build's:

* a causal convolution pads on the left only (the reference pads both sides and
  chops the right end off again);
* time runs along the LAST axis of every attention tensor (`[hyp, T']` rows):
  the location filter is one batched product over sliding windows of the previous
  alignment, the softmax and the context reduce over contiguous rows;
* the training targets (one-hot, smoothed along time, renormalised) and the
  accuracy are built without Python loops over the batch.
"""
from __future__ import absolute_import, division, print_function

import os

import torch
import torch.nn.functional as F
from torch import nn

from att_speech.lm_fst import LmFst
from att_speech.modules.beam_search import (BeamSearch, BeamSearchLM, GraphSearch,
                                             RescoreSearchLM)

_MASKED = -1e5          # additive score of a padded encoder frame


class _CausalConv1d(nn.Conv1d):
    """Dilated 1-D convolution whose output at time t sees inputs <= t."""

    def __init__(self, channels_in, channels_out, taps, dilation):
        super(_CausalConv1d, self).__init__(channels_in, channels_out, taps, dilation=dilation)
        self.history = (taps - 1) * dilation

    def forward(self, x):                       # [batch, channels, time]
        return F.conv1d(F.pad(x, (self.history, 0)), self.weight, self.bias,
                        dilation=self.dilation)


class TemporalBlock(nn.Module):
    """`n_layers` weight-normalised causal convolutions, each followed by ReLU and
    channel dropout, plus a residual connection (1x1 convolution when the channel
    count changes).  Parameters live under `net.conv<j>` / `downsample`."""

    def __init__(self, n_inputs, n_outputs, kernel_size, stride, dilation,
                 padding, dropout=0.2, n_layers=2):
        super(TemporalBlock, self).__init__()
        if stride != 1 or padding != (kernel_size - 1) * dilation:
            raise ValueError("TemporalBlock: only the causal stride-1 form the TCN builds")
        self.net = nn.ModuleDict()
        for j in range(n_layers):
            conv = _CausalConv1d(n_outputs if j else n_inputs, n_outputs, kernel_size, dilation)
            # weight_g / weight_v parametrisation; like the reference, the N(0, 0.01)
            # draw it makes afterwards lands in the derived `.weight` and is overwritten
            # by the next forward, so the effective initialisation is Conv1d's default
            self.net['conv%d' % j] = torch.nn.utils.weight_norm(conv)
        self.channel_dropout = dropout
        self.downsample = None
        if n_inputs != n_outputs:
            self.downsample = nn.Conv1d(n_inputs, n_outputs, 1)
            nn.init.normal_(self.downsample.weight, 0.0, 0.01)

    def forward(self, x):                       # [batch, channels, time]
        y = x
        for conv in self.net.values():
            y = F.dropout1d(torch.relu(conv(y)), self.channel_dropout, self.training)
        shortcut = x if self.downsample is None else self.downsample(x)
        return torch.relu(y + shortcut)


class TCN(nn.Module):
    """Stack of TemporalBlocks, one per entry of `dilation_sizes`."""

    def __init__(self, num_inputs, num_channels, dilation_sizes, kernel_size=2,
                 dropout=0.2, layers_per_block=2):
        super(TCN, self).__init__()
        if len(num_channels) != len(dilation_sizes):
            raise ValueError('num_channels and dilations_sizes lengths '
                             'must be equal (number of blocks)')
        # label history the decoder feeds per step (reference :93)
        self.eff_history = 1 + (kernel_size - 1) * sum(dilation_sizes)
        widths = [num_inputs] + list(num_channels)
        self.network = nn.Sequential(*[
            TemporalBlock(widths[i], widths[i + 1], kernel_size, stride=1, dilation=d,
                          padding=(kernel_size - 1) * d, dropout=dropout,
                          n_layers=layers_per_block)
            for i, d in enumerate(dilation_sizes)])

    def forward(self, x):                       # [time, batch, channels] in and out
        return self.network(x.permute(1, 2, 0)).permute(2, 0, 1)

    def last_step_plan(self, steps):
        """Operands for `last_step` over a window of `steps` frames: per block the frames its
        output is needed at, and per convolution (weight [taps*Cin, Cout] with the
        weight-norm already applied, bias, tap source indices).  Only what the LAST output
        frame depends on is computed — 18 of the 28 frame x layer pairs for the tcn.yaml
        stack — as dense products over all hypotheses."""
        blocks = list(self.network)
        need = [None] * (len(blocks) + 1)
        need[-1] = [steps - 1]
        plans = [None] * len(blocks)
        for bi in range(len(blocks) - 1, -1, -1):
            convs = list(blocks[bi].net.values())
            outs, cur = [], need[bi + 1]
            for conv in reversed(convs):
                taps, dil = conv.kernel_size[0], conv.dilation[0]
                src = sorted({t - (taps - 1 - j) * dil for t in cur for j in range(taps)} - set(
                    range(-steps * taps * dil, 0)))
                outs.append((conv, cur, src))
                cur = src
            need[bi] = sorted(set(cur) | set(need[bi + 1]))          # + the residual input
            layers, avail = [], need[bi]
            for conv, out_pos, _ in reversed(outs):
                taps, dil = conv.kernel_size[0], conv.dilation[0]
                w = torch._weight_norm(conv.weight_v, conv.weight_g, 0).detach()
                # row index into [zero frame] + available frames, per (output frame, tap)
                idx = [[(avail.index(t - (taps - 1 - j) * dil) + 1) if t - (taps - 1 - j) * dil >= 0 else 0
                        for j in range(taps)] for t in out_pos]
                layers.append((w.permute(2, 1, 0).reshape(-1, w.size(0)).contiguous(),
                               conv.bias.detach(), torch.tensor(idx, device=w.device)))
                avail = out_pos
            res_idx = torch.tensor([need[bi].index(t) for t in need[bi + 1]], device=w.device)
            plans[bi] = (layers, res_idx, blocks[bi].downsample)
        return need[0], plans

    @staticmethod
    def last_step(x, plan):
        """x [steps, hyp, C] -> LM state of the last frame [hyp, C] (eval mode)."""
        first_need, plans = plan
        cur = x[first_need] if len(first_need) != x.size(0) else x        # [n, hyp, C]
        for layers, res_idx, downsample in plans:
            y = cur
            for weight, bias, idx in layers:
                padded = torch.cat((torch.zeros_like(y[:1]), y))            # frame 0 = zeros
                cols = padded[idx]                                          # [out, taps, hyp, C]
                n_out, taps, hyps, ch = cols.shape
                cols = cols.permute(0, 2, 1, 3).reshape(n_out * hyps, taps * ch)
                y = torch.relu(torch.addmm(bias, cols, weight)).view(n_out, hyps, -1)
            shortcut = cur[res_idx]
            if downsample is not None:
                shortcut = downsample(shortcut.permute(1, 2, 0)).permute(2, 0, 1)
            cur = torch.relu(y + shortcut)
        return cur[-1]


class LocalAttention(nn.Module):
    """Location-aware additive attention: the score of encoder frame t for a
    hypothesis is  w . tanh(E_t + (a_prev * k)(t) + g)  with E the projected encoder
    output, k a per-hypothesis filter predicted from the LM state that is slid
    causally over the previous alignment a_prev, and g a global LM term."""

    def __init__(self, encoded_size, lm_state_size, hidden_size, kernel_size=32,
                 temperature=1.0, force_forward=None, learnable_init=True, **kwargs):
        super(LocalAttention, self).__init__(**kwargs)
        self.encoded_size, self.kernel_size = encoded_size, kernel_size
        self.temperature, self.force_forward = temperature, force_forward
        self.learnable_init = learnable_init
        self.encoded_to_hidden = nn.Linear(encoded_size, hidden_size)
        self.hidden_to_score = nn.Linear(hidden_size, 1)
        nn.init.zeros_(self.hidden_to_score.weight)     # start from a uniform alignment
        self.lm_to_kernel = nn.Linear(lm_state_size, kernel_size * hidden_size)
        self.lm_to_global = nn.Linear(lm_state_size, hidden_size)
        self.encoded_to_init_weights = nn.Linear(encoded_size, 1)

    @staticmethod
    def _padding_scores(lens, steps, device):
        """[T', B]: 0 on frames of the utterance, -1e5 behind its end."""
        lens = torch.as_tensor(lens).to(device).long()
        behind = torch.arange(steps, device=device)[:, None] >= lens[None, :]
        return behind.float() * _MASKED

    def init_attention(self, encoded, encoded_lens):
        """encoded [T', B, E] -> ((projected encoder, padding scores), alignment [T', B])"""
        pad = self._padding_scores(encoded_lens, encoded.size(0), encoded.device)
        if self.learnable_init:
            first = F.softmax(self.encoded_to_init_weights(encoded).squeeze(2) + pad, 0)
        else:                                   # all mass on the first frame
            first = torch.zeros_like(pad)
            first[0] = 1.0
        return (self.encoded_to_hidden(encoded), pad), first

    def recompute_forward_mask(self, prev_rows, pad):
        """`force_forward = (lo, hi)`: frames outside [peak+lo, peak+hi) of the previous
        alignment are masked, unless that alignment is diffuse (peak < 0.1).
        prev_rows [hyp, T'], pad [T', hyp]."""
        peak, where = prev_rows.max(1)
        t = torch.arange(pad.size(0), device=pad.device)[:, None]
        lo, hi = where + self.force_forward[0], where + self.force_forward[1]
        outside = ((t < lo[None, :]) | (t >= hi[None, :])) & (peak >= 0.1)[None, :]
        return pad + outside.to(pad.dtype) * _MASKED

    def scores(self, att_state, lm_state, prev_att_weights):
        """Masked, temperature-scaled scores as rows [hyp, T']."""
        projected, pad = att_state                               # [T', hyp, A], [T', hyp]
        steps, hyps, width = projected.shape
        prev_rows = prev_att_weights.t()                         # [hyp, T']
        taps = self.kernel_size
        # (a_prev * k)(t) = sum_j a_prev[t - (taps-1) + j] k[j]: a batched product over
        # sliding windows of the left-padded previous alignment
        windows = F.pad(prev_rows, (taps - 1, 0)).unfold(1, taps, 1)            # [hyp, T', taps]
        filters = self.lm_to_kernel(lm_state).view(hyps, width, taps)
        moved = torch.bmm(windows, filters.transpose(1, 2))                      # [hyp, T', A]
        hidden = projected.transpose(0, 1) + moved + self.lm_to_global(lm_state)[:, None, :]
        energy = self.hidden_to_score(torch.tanh(hidden)).squeeze(2) * self.temperature
        if self.force_forward:
            pad = self.recompute_forward_mask(prev_rows, pad)
        return energy + pad.t()

    def forward(self, att_state, lm_state, prev_att_weights):
        """lm_state [hyp, H], prev_att_weights [T', hyp] -> (att_state, alignment [T', hyp])"""
        rows = F.softmax(self.scores(att_state, lm_state, prev_att_weights), 1)
        return att_state, rows.t()


_SMOOTHING_TAPS = (0.005, 0.02, 0.95, 0.02, 0.005)      # along the label axis (:411-414)


class AttentionDecoderTCN(nn.Module):
    def __init__(self, sample_batch, num_classes, tcn_hidden_size, att_hidden_size,
                 dropout_p, learnable_initial_attention=True, label_smoothing=True,
                 kernel_size=3, dilation_sizes=[1, 2, 4], coverage_tau=0.5,
                 coverage_weight=0, beam_size=1, length_normalization=0.0,
                 att_force_forward=None, vocabulary=None, branching_threshold=0.0,
                 lm_file=None, lm_weight=1.0, attention_temperature=1.0,
                 tcn_layers_per_block=2, min_attention_pos=0.5, keep_eos_score=False,
                 use_graph_search=False, graph_search_history_len=-1,
                 graph_search_merge_threshold=0.8, **kwargs):
        super(AttentionDecoderTCN, self).__init__(**kwargs)
        # search options
        self.beam_size, self.length_normalization = beam_size, length_normalization
        self.branching_threshold = branching_threshold
        self.coverage_tau, self.coverage_weight = coverage_tau, coverage_weight
        self.min_attention_pos, self.keep_eos_score = min_attention_pos, keep_eos_score
        self.use_graph_search = use_graph_search
        self.graph_search_history_len = graph_search_history_len
        self.graph_search_merge_threshold = graph_search_merge_threshold
        self.lm_weight, self.rescore = lm_weight, None
        self.TRANSCRIPTION_LEN_GUARD = 250
        # sizes; the class inventory gets an end-of-sequence symbol behind the last class
        self.encoded_size = sample_batch["features"].size(2)
        self.tcn_hidden_size, self.att_hidden_size = tcn_hidden_size, att_hidden_size
        self.EOS, self.num_classes = num_classes, num_classes + 1
        self.vocabulary, self.label_smoothing = vocabulary, label_smoothing
        # modules (attribute names are checkpoint keys)
        self.embedding = nn.Embedding(self.num_classes, tcn_hidden_size)
        self.dropout = nn.Dropout(dropout_p)
        self.attn = LocalAttention(self.encoded_size, tcn_hidden_size, att_hidden_size,
                                   temperature=attention_temperature,
                                   learnable_init=learnable_initial_attention,
                                   force_forward=att_force_forward)
        self.tcn = TCN(tcn_hidden_size, [tcn_hidden_size] * len(dilation_sizes),
                       dilation_sizes=dilation_sizes, kernel_size=kernel_size,
                       dropout=dropout_p, layers_per_block=tcn_layers_per_block)
        width = 256
        self.combined_to_output = nn.Sequential(
            nn.Linear(tcn_hidden_size + self.encoded_size, width), nn.ReLU(), nn.Dropout(dropout_p),
            nn.Linear(width, width), nn.ReLU(), nn.Dropout(dropout_p))
        self.output_to_logits = nn.Linear(width, self.num_classes)
        # language model for the fused searches (the reference reads a pywrapfst.Fst, :293-300)
        self.lm = None
        if lm_file:
            assert vocabulary is not None
            self.lm = lm_file if isinstance(lm_file, LmFst) else LmFst.read(lm_file)
        self.alphabet_mapping = self.create_alphabet_mapping()

    def create_alphabet_mapping(self):
        """LM input label of every model class (:306-327): by symbol name, the space as
        '<spc>'; classes the LM does not know — and EOS — also map to '<spc>'."""
        if self.lm is None:
            return None
        label_of = {sym: lab for lab, sym in self.lm.input_symbols()}
        names = ['<spc>' if s == ' ' else s for s in list(self.vocabulary) + ['<eos>']]
        return [label_of.get(name, label_of['<spc>']) for name in names]

    def hash_dec(self, decoded):
        """Merge key of GraphSearch (:335-345): the last `history` labels, left-filled with -1."""
        span = self.graph_search_history_len
        if span < 0:
            span = self.tcn.eff_history
        if span == 0:
            return 0
        tail = decoded[-span:].tolist()
        return hash(tuple([-1] * (span - len(tail)) + tail))

    # ---------------------------------------------------------------- training
    def _step_output(self, lm_state, context):
        return self.output_to_logits(self.combined_to_output(torch.cat((lm_state, context), -1)))

    def _smoothed_targets(self, labels):
        """labels [B, L] -> per-position target distributions [B, L, classes]: one-hot,
        optionally smeared over NEIGHBOURING POSITIONS with the 5-tap kernel, renormalised,
        class 0 (padding) removed afterwards (:404-424)."""
        onehot = F.one_hot(labels, self.num_classes).to(torch.float32)        # [B, L, C]
        if self.label_smoothing:
            B, L, C = onehot.shape
            taps = onehot.new_tensor(_SMOOTHING_TAPS).view(1, 1, -1)
            along_time = onehot.transpose(1, 2).reshape(B * C, 1, L)
            onehot = F.conv1d(along_time, taps, padding=len(_SMOOTHING_TAPS) // 2) \
                .view(B, C, L).transpose(1, 2)
        dist = onehot / onehot.sum(2, keepdim=True)
        dist[:, :, 0] = 0
        return dist

    def forward(self, encoded, encoded_lens, texts, text_lens,
                return_att_weights=False, **kwargs):
        """Teacher-forced loss (:357-440): the TCN reads <start> + labels, every label
        position attends from the previous position's alignment; cross-entropy against
        the smoothed targets over labels + EOS."""
        dev = encoded.device
        B, L = texts.size(0), texts.size(1) + 1
        labels = torch.zeros(B, L, dtype=torch.long)
        labels[:, :L - 1] = texts.cpu().long()
        labels[torch.arange(B), torch.as_tensor(text_lens).long()] = self.EOS
        labels = labels.to(dev)
        history = self.embedding(labels.t())                                   # [L, B, D]
        lm_states = self.tcn(torch.cat((torch.zeros_like(history[:1]), history[:-1])))
        att_state, alignment = self.attn.init_attention(encoded, encoded_lens)
        enc_rows = encoded.transpose(0, 1)                                     # [B, T', E]
        alignments, step_logits = [], []
        for lm_state in lm_states:
            att_state, alignment = self.attn(att_state, lm_state, alignment)
            alignments.append(alignment)
            context = torch.bmm(alignment.t().unsqueeze(1), enc_rows).squeeze(1)
            step_logits.append(self._step_output(lm_state, context))
        logits = torch.stack(step_logits, 1)                                   # [B, L, C]
        targets = self._smoothed_targets(labels)
        per_position = -(F.log_softmax(logits, 2) * targets).sum(2)
        loss = per_position.mean() / targets.sum(2).mean()
        # accuracy over non-padding positions
        real = labels != 0
        hits = (logits.argmax(2) == labels) & real
        acc = hits.double().sum() / real.double().sum()
        ret = {'loss': loss, 'acc': acc, 'logits': logits}
        if return_att_weights:
            ret['attweights'] = alignments
        return ret

    # ---------------------------------------------------------------- decoding
    def enc_initial_state(self, encoded, encoded_lens, beam_size, batch_size):
        """Search state for `batch_size * beam_size` hypotheses (hypotheses of one
        utterance adjacent): zero label history, the encoder output and its initial
        alignment repeated per beam entry (:442-463)."""
        hyps = batch_size * beam_size
        per_hyp = encoded.repeat_interleave(beam_size, dim=1)
        lens = torch.as_tensor(encoded_lens).repeat_interleave(self.beam_size)
        att_state, alignment = self.attn.init_attention(per_hyp, lens)
        history = encoded.new_zeros(self.tcn.eff_history, hyps, self.tcn_hidden_size)
        return {'inputs': history, 'encoded': per_hyp, 'att_state': att_state,
                'att_weights': alignment}

    def enc_step(self, inputs, encoded, att_state, att_weights):
        """One label step for every live hypothesis (:465-474): LM state from the last
        `eff_history` embeddings, new alignment, context, class logits [1, hyp, C]."""
        lm_state = self.tcn(inputs)[-1]
        att_state, att_weights = self.attn(att_state, lm_state, att_weights)
        context = torch.bmm(att_weights.t().unsqueeze(1), encoded.transpose(0, 1)).squeeze(1)
        logits = self._step_output(lm_state, context).unsqueeze(0)
        return logits, {'encoded': encoded, 'att_state': att_state, 'att_weights': att_weights}

    def _make_search(self, batch_size, device):
        plain = (batch_size, self.beam_size, device, self.num_classes, self.length_normalization)
        if not self.lm:
            return BeamSearch(*plain)
        fused = (self.lm, self.lm_weight, self.alphabet_mapping, self.min_attention_pos,
                 self.coverage_tau, self.coverage_weight) + plain
        if self.rescore:
            return RescoreSearchLM(self.rescore, *fused, keep_eos_score=self.keep_eos_score)
        if self.use_graph_search:
            return GraphSearch(self.hash_dec, self.graph_search_merge_threshold, *fused,
                               keep_eos_score=self.keep_eos_score)
        return BeamSearchLM(*fused, keep_eos_score=self.keep_eos_score)

    def _native_decode_ok(self, encoded):
        C, beam = self.num_classes, self.beam_size
        if os.environ.get('ASR_TCN_NATIVE', '1') == '0':     # A/B switch: torch ops + BeamSearch
            return False
        return (encoded.is_cuda and not self.lm and not self.training
                and not self.attn.force_forward and self.attn.kernel_size == 32
                and beam <= 32 and beam * (C - 1) <= 2048 and encoded.dtype == torch.float32)

    def _decode_native(self, encoded, encoded_lens, return_attention, poll_every=8):
        """The MI355X decode loop for the plain beam search: per label step the LM state of
        the last frame as dense products (TCN.last_step), ONE launch for the local attention
        + context (asr_tcn_attention_step_f32), the output MLP, ONE launch for the beam
        bookkeeping (asr_beam_step_f32) — no host read-back inside a step; the all-finished
        flag is polled every `poll_every` steps (steps behind the flag change nothing)."""
        from att_speech import _native
        from att_speech.modules.beam_search import DeviceBeamSearch
        T, B, E = encoded.shape
        beam, dev, attn = self.beam_size, encoded.device, self.attn
        hyps = B * beam
        lens = torch.as_tensor(encoded_lens).to(dev, torch.int32)
        search = DeviceBeamSearch(B, beam, dev, self.num_classes, self.length_normalization,
                                  self.TRANSCRIPTION_LEN_GUARD)
        # per-utterance operands (the reference repeats them per hypothesis, :449-456)
        (eproj, _), first = attn.init_attention(encoded, lens)
        eproj = eproj.contiguous()
        enc = encoded.contiguous()
        att = first.t().repeat_interleave(beam, dim=0).contiguous()          # [hyp, T]
        plan = self.tcn.last_step_plan(self.tcn.eff_history)
        w_att = torch.cat((attn.lm_to_kernel.weight, attn.lm_to_global.weight)).detach()
        b_att = torch.cat((attn.lm_to_kernel.bias, attn.lm_to_global.bias)).detach()
        n_filt = attn.lm_to_kernel.out_features
        w_score = attn.hidden_to_score.weight.detach().reshape(-1).contiguous()
        b_score = float(attn.hidden_to_score.bias)
        history = enc.new_zeros(self.tcn.eff_history, hyps, self.tcn_hidden_size)
        parent = None
        trace_att = [first.repeat_interleave(beam, dim=1).detach()] if return_attention else None
        trace_logits = []
        for step in range(self.TRANSCRIPTION_LEN_GUARD):
            lm_state = TCN.last_step(history, plan)
            fg = torch.addmm(b_att, lm_state, w_att.t())
            att, context = _native.tcn_attention_step(
                eproj, enc, lens, fg[:, :n_filt].contiguous(), fg[:, n_filt:].contiguous(),
                w_score, b_score, attn.temperature, att, parent, beam)
            logits = self._step_output(lm_state, context)
            chosen, parent = search.step(logits)
            if return_attention:
                trace_logits.append(logits.detach()[None])
                trace_att.append(att.t().detach())
            history = torch.cat((history[1:].index_select(1, parent.long()),
                                 self.embedding(chosen.long())[None]))
            if (return_attention or step % poll_every == poll_every - 1) and search.poll_finished():
                break
        search.finalize()
        out = {'decoded': search.best_finished,
               'decoded_scores': search.best_finished_scores_elements,
               'loss': torch.Tensor(search.best_finished_scores).mean()}
        if return_attention:
            out.update(attweights=trace_att, logits=trace_logits)
        out.update(coverage=None, graph=None, beam_search=search)
        return out

    def decode(self, encoded, encoded_lens, texts=None, text_lens=None,
               return_attention=False, print_debug=False, **kwargs):
        """Beam search over label steps (:476-585), at most TRANSCRIPTION_LEN_GUARD of them."""
        if self._native_decode_ok(encoded) and not print_debug:
            return self._decode_native(encoded, encoded_lens, return_attention)
        search = self._make_search(encoded.size(1), encoded.device)
        search.print_debug = print_debug
        state = self.enc_initial_state(encoded, encoded_lens, self.beam_size, encoded.size(1))
        trace_att = [state['att_weights'].detach()] if return_attention else None
        trace_logits = []
        for _ in range(self.TRANSCRIPTION_LEN_GUARD):
            history = state['inputs']
            logits, state = self.enc_step(**state)
            if return_attention:
                trace_logits.append(logits.detach())
                trace_att.append(state['att_weights'].detach())
            chosen, parent = search.step(logits, att_weights=state['att_weights'])
            # survivors inherit their parent's alignment and label history
            state['att_weights'] = state['att_weights'][:, parent]
            state['inputs'] = torch.cat((history[1:, parent], self.embedding(chosen)[None]))
            if search.has_finished():
                break
        out = {'decoded': search.best_finished,
               'decoded_scores': search.best_finished_scores_elements,
               'loss': torch.Tensor(search.best_finished_scores).mean()}
        if return_attention:
            out.update(attweights=trace_att, logits=trace_logits)
        out.update(coverage=search.coverage, graph=search.get_graph(), beam_search=search)
        return out
