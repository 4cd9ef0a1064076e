"""Statement-level overlap of the host mirrors with the same-named reference files
(development check for the copy rule: comment/blank-stripped lines, difflib matching
blocks).  Needs /root/reference; prints the share of OUR statements found in matching
blocks of >= `--min-block` consecutive statements."""
import argparse
import difflib
import io
import os
import sys
import tokenize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [
    ('pytorch-asr_amd/att_speech/modules/decoders/advanced_decoder.py', 'att_speech/modules/decoders/advanced_decoder.py'),
    ('pytorch-asr_amd/att_speech/modules/tcn.py', 'att_speech/modules/tcn.py'),
    ('pytorch-asr_amd/att_speech/modules/encoders/encoder_utils.py', 'att_speech/modules/encoders/encoder_utils.py'),
    ('pytorch-asr_amd/att_speech/modules/encoders/deep_speech_2.py', 'att_speech/modules/encoders/deep_speech_2.py'),
    ('pytorch-asr_amd/att_speech/modules/hooks/gradient_clipping.py', 'att_speech/modules/hooks/gradient_clipping.py'),
    ('pytorch-asr_amd/att_speech/modules/hooks/polyak.py', 'att_speech/modules/hooks/polyak.py'),
    ('pytorch-asr_amd/att_speech/models.py', 'att_speech/models.py'),
    ('pytorch-asr_amd/att_speech/utils.py', 'att_speech/utils.py'),
    ('pytorch-asr_amd/att_speech/fst_utils.py', 'att_speech/fst_utils.py'),
    ('pytorch-asr_amd/att_speech/modules/beam_search.py', 'att_speech/modules/beam_search.py'),
    ('pytorch-asr_amd/att_speech/modules/ctc_losses.py', 'att_speech/modules/ctc_losses.py'),
    ('pytorch-asr_amd/att_speech/ctc_forward.py', 'ctc_forward.py'),
]


def statements(path):
    """logical lines without comments, docstrings and blank lines, whitespace-normalised"""
    src = open(path, encoding='utf-8', errors='replace').read()
    out, cur, prev = [], [], None
    try:
        for tok in tokenize.generate_tokens(io.StringIO(src).readline):
            if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.INDENT, tokenize.DEDENT, tokenize.ENCODING):
                continue
            if tok.type == tokenize.NEWLINE:
                if cur and not (len(cur) == 1 and prev == tokenize.STRING):
                    out.append(' '.join(cur))
                cur = []
                continue
            cur.append(tok.string)
            prev = tok.type
    except (tokenize.TokenError, IndentationError, SyntaxError):
        out = [' '.join(l.split()) for l in src.splitlines()
               if l.strip() and not l.strip().startswith('#')]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--min-block', type=int, default=1)
    a = ap.parse_args()
    for ours, theirs in PAIRS:
        po, pt = os.path.join(ROOT, ours), os.path.join(a.ref, theirs)
        if not (os.path.exists(po) and os.path.exists(pt)):
            continue
        so, st = statements(po), statements(pt)
        sm = difflib.SequenceMatcher(None, so, st, autojunk=False)
        same = sum(b.size for b in sm.get_matching_blocks() if b.size >= a.min_block)
        print('%5.1f %%  (%3d of %3d statements)  %s' % (100.0 * same / max(1, len(so)), same, len(so), ours))


if __name__ == '__main__':
    sys.exit(main())
