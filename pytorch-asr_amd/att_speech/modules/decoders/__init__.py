from att_speech.modules.decoders.advanced_decoder import (  # noqa: F401
    CTCDecoderAdvanced, FSTDecoder, LutLinear, NGramLinear)

__all__ = ['CTCDecoderAdvanced', 'FSTDecoder', 'LutLinear', 'NGramLinear']
