"""att_speech — MI355X-native implementation of the CTC / lattice training and
decode hot path of chorowski-lab/pytorch-asr behind the reference's own dotted
module paths (att_speech.fst_utils, att_speech.modules.*, att_speech.models)."""
