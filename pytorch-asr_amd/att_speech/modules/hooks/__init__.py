"""reference modules/hooks/__init__.py — the hooks that touch gradients or
parameters on the training step."""
from att_speech.modules.hooks.gradient_clipping import GradientClipping
from att_speech.modules.hooks.hook import TrainingLoopHook
from att_speech.modules.hooks.polyak import PolyakDecay

__all__ = ['GradientClipping', 'PolyakDecay', 'TrainingLoopHook']
