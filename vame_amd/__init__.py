"""vame_amd -- MI355X-native implementation of VAME's RNN-VAE training + latent-embedding path.

`import vame_amd as vame` gives the workflow calls of the reference that sit on this path
(vame/__init__.py:16,18):  vame.train_model(config)  and  vame.pose_segmentation(config),  plus the two that reuse
its kernels in eval mode (:17,23):  vame.evaluate_model(config)  and  vame.generative_model(config, mode),  and the step that
writes the training set the path reads (:15):  vame.create_trainset(config).
"""
from .analysis.generative_functions import generative_model  # noqa: F401
from .analysis.pose_segmentation import pose_segmentation  # noqa: F401
from .model.create_training import create_trainset  # noqa: F401
from .model.evaluate import evaluate_model  # noqa: F401
from .model.rnn_vae import train_model  # noqa: F401

__version__ = "0.1.0"
