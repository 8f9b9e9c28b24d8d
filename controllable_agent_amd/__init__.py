"""controllable_agent_amd -- MI355X (gfx950) implementation of the FB-DDPG update hot path of
facebookresearch/controllable_agent behind the reference's Agent / ReplayBuffer plugin surface.

    from controllable_agent_amd import FBHipAgent, DeviceReplayBuffer

Compute goes through libfbhip.so (hand-written HIP kernels, include/fbhip.h); there is no CPU fallback.
"""
from ._lib import LIB_PATH  # noqa: F401

__all__ = ["FBHipAgent", "FBDDPGAgentConfig", "DiscreteFBHipAgent", "DiscreteFBAgentConfig", "SFHipAgent", "SFAgentConfig",
           "DeviceReplayBuffer", "EpisodeBatch", "kernels"]


def __getattr__(name):
    if name in ("FBHipAgent", "FBDDPGAgentConfig", "DiscreteFBHipAgent", "DiscreteFBAgentConfig", "SFHipAgent", "SFAgentConfig"):
        from . import agent
        return getattr(agent, name)
    if name in ("DeviceReplayBuffer", "EpisodeBatch"):
        from . import replay
        return getattr(replay, name)
    if name == "kernels":
        import importlib
        return importlib.import_module(".kernels", __name__)
    raise AttributeError(name)
