from .dataloader import SEQUENCE_DATASET  # noqa: F401
from .rnn_vae import train_model  # noqa: F401
from .create_training import create_trainset  # noqa: F401
from .evaluate import evaluate_model  # noqa: F401
