"""border_amd -- MI355X-native opt-step engine behind border's Agent / ReplayBufferBase traits.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of include/border_amd.h)
and the host-side mirror of the reference interfaces (replay.py, dqn.py, trainer.py).
"""
from ._lib import BdrError, device_count  # noqa: F401
from .replay import GenericTransitionBatch, PerConfig, SimpleReplayBuffer, SimpleReplayBufferConfig  # noqa: F401
from .dqn import AtariCnnConfig, Dqn, DqnConfig, DqnModelConfig, EpsilonGreedy, MlpConfig, OptimizerConfig, Softmax  # noqa: F401
from .sac import Sac, SacConfig  # noqa: F401
from .iqn import Iqn, IqnConfig  # noqa: F401
from . import checkpoint  # noqa: F401
from .atari import AtariDeviceEnv, AtariPreprocessor  # noqa: F401
from .trainer import (NativeTrainer, ParamExchange, Sampler, SimpleStepProcessor, Step, SyntheticEnv, Trainer, TrainerConfig,  # noqa: F401
                      shard_seed)
from .async_trainer import (ActorManagerConfig, ActorStat, AsyncTrainer, AsyncTrainerConfig, AsyncTrainStat, ModelMailbox,  # noqa: F401
                            actor_stats_fmt)
