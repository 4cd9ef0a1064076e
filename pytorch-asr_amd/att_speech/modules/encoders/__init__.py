from att_speech.modules.encoders.deep_speech_2 import DeepSpeech2  # noqa: F401

__all__ = ['DeepSpeech2']
