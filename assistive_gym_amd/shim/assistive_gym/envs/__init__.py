"""`assistive_gym.envs`: the env classes by the reference's names (assistive_gym/envs/__init__.py imports them the same way)."""
from assistive_gym_amd.envs import ENV_IDS as _IDS

for _cls in _IDS.values():
    globals()[_cls.__name__] = _cls
__all__ = [c.__name__ for c in _IDS.values()]
