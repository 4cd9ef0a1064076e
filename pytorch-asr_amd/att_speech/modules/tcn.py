"""att_speech.modules.tcn — TCN language-model + local-attention decoder of the
reference (att_speech/modules/tcn.py): Chomp1d (:33-43), TemporalBlock (:46-86),
TCN (:89-116), LocalAttention (:119-230), AttentionDecoderTCN (:233-585) with
its training `forward` (:357-440) and step-wise beam `decode` (:442-585: plain
BeamSearch, or with `lm_file` the LM-fused BeamSearchLM / RescoreSearchLM /
GraphSearch over an att_speech.lm_fst.LmFst).  Same constructor kwargs, return dicts and state_dict keys
(`tcn.network.{i}.net.conv{j}.{bias,weight_g,weight_v}`, ...).

This stage is dense small-GEMM / elementwise work on [B*beam, 384] states; it
runs on stock torch device ops in this round (no hand-written kernels yet)."""
from __future__ import absolute_import, division, print_function

from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

from att_speech.lm_fst import LmFst
from att_speech.modules.beam_search import (BeamSearch, BeamSearchLM, GraphSearch,
                                             RescoreSearchLM)
from att_speech.utils import get_mask


class Chomp1d(nn.Module):
    def __init__(self, chomp_size):
        super(Chomp1d, self).__init__()
        self.chomp_size = chomp_size

    def forward(self, x):            # batch x hidden x seq
        return x[:, :, :-self.chomp_size].contiguous()


class TemporalBlock(nn.Module):
    def __init__(self, n_inputs, n_outputs, kernel_size, stride, dilation,
                 padding, dropout=0.2, n_layers=2):
        super(TemporalBlock, self).__init__()
        layers = []
        for i in range(n_layers):
            n_in = n_inputs if i == 0 else n_outputs
            conv = torch.nn.utils.weight_norm(nn.Conv1d(
                n_in, n_outputs, kernel_size, stride=stride, padding=padding,
                dilation=dilation))
            layers += [('conv' + str(i), conv), ('chomp' + str(i), Chomp1d(padding)),
                       ('relu' + str(i), nn.ReLU()), ('drop' + str(i), nn.Dropout2d(dropout))]
        self.net = nn.Sequential(OrderedDict(layers))
        self.downsample = nn.Conv1d(n_inputs, n_outputs, 1) if n_inputs != n_outputs else None
        self.relu = nn.ReLU()
        self.init_weights()

    def init_weights(self):
        for name, layer in self.net.named_children():
            if name.startswith('conv'):
                layer.weight.data.normal_(0, 0.01)
        if self.downsample is not None:
            self.downsample.weight.data.normal_(0, 0.01)

    def forward(self, x):
        out = self.net(x)
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(out + res)


class TCN(nn.Module):
    def __init__(self, num_inputs, num_channels, dilation_sizes, kernel_size=2,
                 dropout=0.2, layers_per_block=2):
        super(TCN, self).__init__()
        self.eff_history = sum(dilation_sizes) * (kernel_size - 1) + 1
        if len(dilation_sizes) != len(num_channels):
            raise ValueError('num_channels and dilations_sizes lengths '
                             'must be equal (number of blocks)')
        layers = []
        for i, d in enumerate(dilation_sizes):
            layers += [TemporalBlock(
                num_inputs if i == 0 else num_channels[i - 1], num_channels[i],
                kernel_size, stride=1, dilation=d, padding=(kernel_size - 1) * d,
                dropout=dropout, n_layers=layers_per_block)]
        self.network = nn.Sequential(*layers)

    def forward(self, x):            # seq x batch x hidden
        return self.network(x.permute(1, 2, 0)).permute(2, 0, 1)


class LocalAttention(nn.Module):
    def __init__(self, encoded_size, lm_state_size, hidden_size, kernel_size=32,
                 temperature=1.0, force_forward=None, learnable_init=True, **kwargs):
        super(LocalAttention, self).__init__(**kwargs)
        self.encoded_size = encoded_size
        self.kernel_size = kernel_size
        self.temperature = temperature
        self.force_forward = force_forward
        self.encoded_to_hidden = nn.Linear(encoded_size, hidden_size)
        self.hidden_to_score = nn.Linear(hidden_size, 1)
        self.lm_to_kernel = nn.Linear(lm_state_size, kernel_size * hidden_size)
        self.lm_to_global = nn.Linear(lm_state_size, hidden_size)
        self.hidden_to_score.weight.data.zero_()     # initially: average everything
        self.encoded_to_init_weights = nn.Linear(encoded_size, 1)
        self.learnable_init = learnable_init

    def init_attention(self, encoded, encoded_lens):
        """(:143-165) encoded [T,B,H] -> ((encoded contribution, mask), weights [T,B])"""
        encoded_contribution = self.encoded_to_hidden(encoded)
        mask = get_mask(encoded_lens, encoded.size(0), batch_first=False)
        mask = ((mask - 1.0) * 1e5).to(encoded.device)
        scores = self.encoded_to_init_weights(encoded).squeeze(2) + mask
        if self.learnable_init:
            att_weights = F.softmax(scores, 0)
        else:
            att_weights = torch.zeros_like(scores)
            att_weights[0, :] = 1
        return (encoded_contribution, mask), att_weights

    def recompute_forward_mask(self, prev_att_weights, mask):
        """(:167-191) prev_att_weights [1,B,T]"""
        att_max, max_ind = torch.max(prev_att_weights, 2)
        att_max, max_ind = att_max.view(-1).tolist(), max_ind.view(-1).tolist()
        mask = mask.clone()
        for j, ind in enumerate(max_ind):
            if att_max[j] < 0.1:          # diffused attention, don't mask
                continue
            left, right = ind + self.force_forward[0], ind + self.force_forward[1]
            if left > 0:
                mask[:left, j] -= 1e5
            if right < mask.shape[0]:
                mask[right:, j] -= 1e5
        return mask

    def forward(self, att_state, lm_state, prev_att_weights):
        """(:193-230) lm_state [B,H], prev_att_weights [T,B] -> weights [T,B]"""
        encoded_contribution, mask = att_state
        # 1: move the previous attention with a per-hypothesis 1-D convolution
        kernel = self.lm_to_kernel(lm_state)
        bs = kernel.size(0)
        kernel = kernel.view(-1, 1, self.kernel_size)
        prev = prev_att_weights.t().unsqueeze(0)
        pad = self.kernel_size - 1
        if prev.is_cuda:
            # the per-hypothesis 1-D convolution (groups = hypotheses, a predicted kernel
            # each) as ONE batched GEMM over a strided window view of the padded previous
            # attention: out[b, t, c] = sum_k prev[b, t + k - pad] * kernel[b, c, k]
            # (MIOpen runs the grouped form as a generic implicit GEMM: 0.32 ms per step)
            T_enc = prev_att_weights.size(0)
            padded = F.pad(prev_att_weights.t(), (pad, 0))                       # [bs, T + pad]
            windows = padded.as_strided((bs, T_enc, self.kernel_size),
                                        (padded.stride(0), 1, 1))                # [bs, T, K]
            kern = kernel.view(bs, -1, self.kernel_size)                          # [bs, C, K]
            local_hidden = torch.bmm(windows, kern.transpose(1, 2))              # [bs, T, C]
            local_hidden = local_hidden.transpose(0, 1).reshape(encoded_contribution.shape)
        else:
            local_hidden = F.conv1d(prev, kernel, padding=pad, groups=bs)[:, :, :-pad]
            local_hidden = local_hidden.transpose(0, 2).reshape(encoded_contribution.shape)
        # 2: match the LM state with the encoded sequence globally; 3: combine
        global_hidden = self.lm_to_global(lm_state).unsqueeze(0)
        hidden = encoded_contribution + local_hidden + global_hidden
        scores = self.hidden_to_score(torch.tanh(hidden)).squeeze(2) * self.temperature
        if self.force_forward:
            mask = self.recompute_forward_mask(prev, mask)
        if scores.is_cuda:      # normalise over time on contiguous rows (softmax over a strided
            return att_state, F.softmax((scores + mask).t().contiguous(), 1).t()   # dim is slow)
        return att_state, F.softmax(scores + mask, 0)


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
        self.rescore = None
        self.use_graph_search = use_graph_search
        self.min_attention_pos = min_attention_pos
        self.keep_eos_score = keep_eos_score
        self.graph_search_history_len = graph_search_history_len
        self.graph_search_merge_threshold = graph_search_merge_threshold
        self.coverage_tau = coverage_tau
        self.coverage_weight = coverage_weight
        self.encoded_size = sample_batch["features"].size()[2]
        self.tcn_hidden_size = tcn_hidden_size
        self.att_hidden_size = att_hidden_size
        self.num_classes = num_classes + 1          # adding EOS
        self.EOS = num_classes
        self.vocabulary = vocabulary
        self.embedding = nn.Embedding(self.num_classes, tcn_hidden_size)
        self.dropout = nn.Dropout(dropout_p)
        self.attn = LocalAttention(self.encoded_size, tcn_hidden_size, att_hidden_size,
                                   temperature=attention_temperature,
                                   learnable_init=learnable_initial_attention,
                                   force_forward=att_force_forward)
        self.tcn = TCN(tcn_hidden_size, [tcn_hidden_size] * len(dilation_sizes),
                       dilation_sizes=dilation_sizes, kernel_size=kernel_size,
                       dropout=dropout_p, layers_per_block=tcn_layers_per_block)
        out_size = 256
        self.combined_to_output = nn.Sequential(
            nn.Linear(self.tcn_hidden_size + self.encoded_size, out_size), nn.ReLU(),
            nn.Dropout(p=dropout_p), nn.Linear(out_size, out_size), nn.ReLU(),
            nn.Dropout(p=dropout_p))
        self.output_to_logits = nn.Linear(out_size, self.num_classes)
        self.beam_size = beam_size
        self.length_normalization = length_normalization
        self.branching_threshold = branching_threshold
        self.TRANSCRIPTION_LEN_GUARD = 250
        self.lm_weight = lm_weight
        self.label_smoothing = label_smoothing
        if lm_file:
            # (:293-300) the reference reads a pywrapfst.Fst; arcs of an LmFst are
            # always input-label sorted
            self.lm = lm_file if isinstance(lm_file, LmFst) else LmFst.read(lm_file)
            assert self.vocabulary is not None
        else:
            self.lm = None
        self.alphabet_mapping = self.create_alphabet_mapping()

    def create_alphabet_mapping(self):
        """model class id -> LM input label (:306-327); symbols the LM does not know
        (and EOS) map to its <spc>."""
        if self.lm is None:
            return None
        lm_ids, lm_syms = zip(*list(self.lm.input_symbols()))
        default_id = lm_ids[lm_syms.index('<spc>')]
        mapping = []
        for s in list(self.vocabulary) + ['<eos>']:
            if s == ' ':
                s = '<spc>'
            mapping.append(lm_ids[lm_syms.index(s)] if s in lm_syms else default_id)
        return mapping

    def hash_dec(self, decoded):
        """hash of the last `history` letters of a hypothesis (:335-345)"""
        hs = (self.graph_search_history_len if self.graph_search_history_len >= 0
              else self.tcn.eff_history)
        if hs == 0:
            return 0
        return hash(tuple([-1] * (hs - len(decoded)) + decoded[-hs:].tolist()))

    def forward(self, encoded, encoded_lens, texts, text_lens,
                return_att_weights=False, **kwargs):
        """Training loss (:357-440): teacher-forced TCN over the label sequence,
        per-label local attention, label-smoothed cross-entropy."""
        bs = texts.size(0)
        dev = encoded.device
        att_state, att_weights = self.attn.init_attention(encoded, encoded_lens)
        texts = torch.cat((texts.cpu().int(), torch.zeros(bs, 1).int()), dim=1)
        for b in range(bs):
            texts[b, int(text_lens[b])] = self.EOS
        max_text_len = texts.size(1)
        texts = texts.long().to(dev)
        embedded = self.embedding(texts.t())                         # L x B x D
        lm_outputs = self.tcn(torch.cat((
            torch.zeros(1, embedded.size(1), embedded.size(2)).type_as(embedded),
            embedded[:-1])))
        all_att_weights, outputs = [], []
        for lm_output in lm_outputs:
            att_state, att_weights = self.attn(att_state, lm_output, att_weights)
            all_att_weights.append(att_weights)
            context = (att_weights.unsqueeze(2) * encoded).sum(0)
            outputs.append(self.combined_to_output(
                torch.cat((lm_output, context), 1)).unsqueeze(0))
        logits = self.output_to_logits(torch.cat(outputs)).permute(1, 0, 2).contiguous()
        targets = torch.zeros(bs, self.num_classes, max_text_len, device=dev)
        targets.scatter_(1, texts.unsqueeze(1), 1)
        if self.label_smoothing:                                      # (:411-420)
            smooth = torch.tensor([0.005, 0.02, 0.95, 0.02, 0.005], device=dev).view(1, 1, -1)
            targets = (F.conv1d(targets.view(bs * self.num_classes, 1, max_text_len),
                                smooth, padding=2)
                       .view(bs, self.num_classes, max_text_len).transpose(1, 2))
        else:
            targets = targets.transpose(1, 2)
        targets = targets / targets.sum(2).unsqueeze(2)
        targets[:, :, 0] = 0                                          # ignore index 0
        loss = (-(F.log_softmax(logits, 2) * targets).sum(2).mean() / targets.sum(2).mean())
        predictions = torch.argmax(logits, dim=2)
        predictions = torch.where(texts == 0, texts, predictions)
        acc = (((predictions == texts).double() - (texts == 0).double()).mean()
               * (torch.ones_like(texts).sum().item() / texts.nonzero().size(0)))
        ret = {'loss': loss, 'acc': acc, 'logits': logits}
        if return_att_weights:
            ret['attweights'] = all_att_weights
        return ret

    def enc_initial_state(self, encoded, encoded_lens, beam_size, batch_size):
        """(:442-463)"""
        max_encoded_len = encoded.size(0)
        inputs = torch.zeros(self.tcn.eff_history, batch_size * beam_size,
                             self.tcn_hidden_size, device=encoded.device)
        encoded = encoded.unsqueeze(2).repeat(1, 1, beam_size, 1) \
            .view(max_encoded_len, batch_size * beam_size, -1)
        ext_lens = torch.as_tensor(encoded_lens).clone().unsqueeze(1) \
            .repeat(1, self.beam_size).view(-1)
        att_state, att_weights = self.attn.init_attention(encoded, ext_lens)
        return {'inputs': inputs, 'encoded': encoded, 'att_state': att_state,
                'att_weights': att_weights}

    def enc_step(self, inputs, encoded, att_state, att_weights):
        """(:465-474) one decoder step for every live hypothesis."""
        lm_output = self.tcn(inputs)[-1]
        att_state, att_weights = self.attn(att_state, lm_output, att_weights)
        if encoded.is_cuda:     # [B,1,T] x [B,T,H] batched product: reads `encoded` once
            context = torch.bmm(att_weights.t().unsqueeze(1), encoded.transpose(0, 1)).squeeze(1)
        else:
            context = (att_weights.unsqueeze(2) * encoded).sum(0)
        combined = torch.cat((lm_output, context), 1).unsqueeze(0)
        logits = self.output_to_logits(self.combined_to_output(combined))
        return logits, {'encoded': encoded, 'att_state': att_state,
                        'att_weights': att_weights}

    def decode(self, encoded, encoded_lens, texts=None, text_lens=None,
               return_attention=False, print_debug=False, **kwargs):
        """(:476-585) beam search, at most TRANSCRIPTION_LEN_GUARD steps."""
        batch_size, beam_size = encoded.size(1), self.beam_size
        base_args = (batch_size, beam_size, encoded.device, self.num_classes,
                     self.length_normalization)
        if self.lm:
            lm_args = (self.lm, self.lm_weight, self.alphabet_mapping, self.min_attention_pos,
                       self.coverage_tau, self.coverage_weight) + base_args
            if self.rescore:
                beam_search = RescoreSearchLM(self.rescore, *lm_args,
                                              keep_eos_score=self.keep_eos_score)
            elif self.use_graph_search:
                beam_search = GraphSearch(self.hash_dec, self.graph_search_merge_threshold,
                                          *lm_args, keep_eos_score=self.keep_eos_score)
            else:
                beam_search = BeamSearchLM(*lm_args, keep_eos_score=self.keep_eos_score)
        else:
            beam_search = BeamSearch(*base_args)
        beam_search.print_debug = print_debug
        enc_state = self.enc_initial_state(encoded, encoded_lens, beam_size, batch_size)
        all_att_weights, all_logits = [], []
        if return_attention:
            all_att_weights += [enc_state['att_weights'].detach()]
        for _ in range(self.TRANSCRIPTION_LEN_GUARD):
            prev_inputs = enc_state['inputs']
            logits, enc_state = self.enc_step(**enc_state)
            if return_attention:
                all_logits += [logits.detach()]
                all_att_weights += [enc_state['att_weights'].detach()]
            new_input, state_mapping = beam_search.step(
                logits, att_weights=enc_state['att_weights'])
            enc_state['att_weights'] = enc_state['att_weights'][:, state_mapping]
            prev_inputs = prev_inputs[:, state_mapping]
            enc_state['inputs'] = torch.cat(
                (prev_inputs[1:], self.embedding(new_input).unsqueeze(0)))
            if beam_search.has_finished():
                break
        results = {'decoded': beam_search.best_finished,
                   'decoded_scores': beam_search.best_finished_scores_elements,
                   'loss': torch.Tensor(beam_search.best_finished_scores).mean()}
        if return_attention:
            results['attweights'] = all_att_weights
            results['logits'] = all_logits
        results['coverage'] = beam_search.coverage
        results['graph'] = beam_search.get_graph()
        results['beam_search'] = beam_search
        return results
