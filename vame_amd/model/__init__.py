from .dataloader import SEQUENCE_DATASET  # noqa: F401
from .rnn_vae import train_model  # noqa: F401
