from .base_env import BaseEnv, TASK_LIST  # noqa: F401
from .base_env_mp import BaseEnvMp  # noqa: F401
