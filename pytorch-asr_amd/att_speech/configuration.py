"""att_speech.configuration — only the process-wide device flag of the
reference (configuration.py:113-115); YAML parsing is out of scope."""


class Globals(object):
    cuda = True
