"""Host-side helpers with the reference's surface (lvmhelpers/utils.py:5-99)."""
import torch.nn as nn


class DeterministicWrapper(nn.Module):
    """Pass-through wrapper: ``forward`` is the wrapped agent's ``forward`` (utils.py:5-17)."""

    def __init__(self, agent):
        super().__init__()
        self.agent = agent

    def forward(self, *args, **kwargs):
        return self.agent(*args, **kwargs)


# (flag, kwargs) in the order and with the defaults of lvmhelpers/utils.py:24-97
_COMMON_FLAGS = (
    ("--mode", dict(type=str, default="sfe",
                    choices=["sfe", "nvil", "gs", "marg", "sumsample", "topksparse", "sparsemap"],
                    help="estimator used to train the discrete / structured latent variable")),
    ("--entropy_coeff", dict(type=float, default=1.0, help="weight of the entropy regulariser")),
    ("--latent_size", dict(type=int, default=10, help="number of categories / bits")),
    ("--normalizer", dict(type=str, default="softmax", choices=["softmax", "sparsemax"],
                          help="map from scores to a distribution over categories")),
    ("--topksparse", dict(type=int, default=10, help="k of top-k sparsemax")),
    ("--noinit", dict(action="store_true")),
    ("--budget", dict(type=int, default=0)),
    ("--gs_tau", dict(type=float, default=1.0, help="Gumbel-Softmax temperature")),
    ("--straight_through", dict(action="store_true")),
    ("--temperature_decay", dict(type=float, default=1e-5,
                                 help="Gumbel-Softmax temperature decay constant")),
    ("--temperature_update_freq", dict(type=int, default=1000,
                                       help="steps between temperature decays")),
    ("--baseline_type", dict(type=str, default="runavg", choices=["runavg", "sample"],
                             help="SFE baseline: running average or self-critic sample")),
    ("--topk", dict(type=int, default=1, help="classes summed by the sum&sample estimator")),
    ("--random_seed", dict(type=int, default=42, help="random seed")),
    ("--n_epochs", dict(type=int, default=10, help="training epochs")),
    ("--load_from_checkpoint", dict(type=str, default=None,
                                    help="checkpoint to restore model/trainer/optimizer from")),
    ("--no_cuda", dict(default=False, action="store_true", help="run on the CPU")),
    ("--batch_size", dict(type=int, default=32, help="training batch size")),
    ("--optimizer", dict(type=str, default="adam", choices=["adam", "sgd", "adagrad"],
                         help="optimizer")),
    ("--lr", dict(type=float, default=1e-3, help="learning rate")),
    ("--weight_decay", dict(type=float, default=1e-5, help="L2 regularisation constant")),
)


def populate_common_params(arg_parser):
    """Register the flags shared by the three experiments (same names, types, defaults, choices)."""
    for flag, kw in _COMMON_FLAGS:
        arg_parser.add_argument(flag, **kw)
    return arg_parser
