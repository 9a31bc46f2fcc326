"""Drop-in for the reference's second architecture module ``stylex/stylex_train_new.py`` (selected by
``USE_OLD_ARCHITECTURE = False`` in cli.py:17-22): conditional discriminator (two logits weighted by the classifier
probabilities of the conditioning batch, :887-916), mapping network on ``latent_dim - 2`` dimensions with the
probabilities appended to W (:332-333, :937-946), encoder in its own Adam parameter group at lr 1e-5 (:967-969), one
backward of ``gen + rec + kl`` per encoder micro-step (:1501-1503).

Everything is the engine of ``stylex_train.py`` run with ``new_architecture=True`` — same HIP kernels, same fused
blocks; this module only binds the reference's names to that mode.
"""
from stylex_train import *  # noqa: F401,F403  (same public surface as the default architecture)
from stylex_train import NanException, StylEx as _StylEx, Trainer as _Trainer, latent_to_w as _latent_to_w  # noqa: F401


def latent_to_w(style_vectorizer, latent_descr, probabilities):  # stylex_train_new.py:332-333
    return _latent_to_w(style_vectorizer, latent_descr, probabilities)


class StylEx(_StylEx):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault("conditional", True)
        super().__init__(*args, **kwargs)


class Trainer(_Trainer):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault("new_architecture", True)
        super().__init__(*args, **kwargs)
