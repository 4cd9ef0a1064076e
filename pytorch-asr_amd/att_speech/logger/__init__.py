"""No-op stand-in for the reference's tensorboard/CSV logger singleton
(att_speech/logger/): modules call logger.log_scalar directly
(advanced_decoder.py:52,199) and must tolerate a logger that drops it."""


class DefaultTensorLogger(object):
    _instance = None

    def __new__(cls, *args, **kwargs):
        if cls._instance is None:
            cls._instance = super(DefaultTensorLogger, cls).__new__(cls)
        return cls._instance

    def log_scalar(self, *args, **kwargs):
        pass

    def is_currently_logging(self):
        return False
