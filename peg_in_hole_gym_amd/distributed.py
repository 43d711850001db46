"""Block-sharding of the env batch over the GPUs of one node: one process per GPU, rank r owns the envs
[r*N/G, (r+1)*N/G); envs never interact, so the step itself needs no collective.  The only exchange step is the optional
all-gather of the stacked observation (RCCL over xGMI when the backend is "nccl"; 40 KB per rank at 2048 envs, i.e.
latency-bound), for learners that want the full batch on every rank."""
import numpy as np
import torch
import torch.distributed as dist

from .envs.utils import env_offsets


class ShardedVecEnv:
    def __init__(self, total_envs, offset=(0, 0, 0), gather_obs=True, backend_factory=None, device=None, **cfg):
        self._dist = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self._dist else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        if total_envs % self.world:
            raise ValueError("total_envs (%d) must be a multiple of the world size (%d)" % (total_envs, self.world))
        self.total = total_envs
        self.n_local = total_envs // self.world
        self.env0 = self.rank * self.n_local
        offs = env_offsets(offset, total_envs)[self.env0:self.env0 + self.n_local]
        if backend_factory is None:
            from .vec_env import PihVecEnv
            self.backend = PihVecEnv(self.n_local, device=device or "cuda:%d" % (self.rank % max(torch.cuda.device_count(), 1)),
                                     offsets=offs, env_index0=self.env0, **cfg)
        else:
            self.backend = backend_factory(self.n_local, offs, env_index0=self.env0, **cfg)
        # gather_obs: True = all-gather when there is more than one rank; 'always' = also at world size 1 (a process group must be
        # initialised): the collective path of BASELINE configs[3] exercised on a single GPU; False = obs stay sharded
        if gather_obs == "always" and not self._dist:
            raise ValueError("gather_obs='always' needs an initialised torch.distributed process group")
        self.gather_obs = bool(gather_obs) and (self.world > 1 or gather_obs == "always")
        self._gathered = None

    def reset(self, mask=None):
        self.backend.reset(mask)

    def step(self, local_actions):
        """local_actions: [n_local,4] for this rank's block.  Returns (obs, reward, done) of the block; when gather_obs
        is on, `self.obs_all` holds the stacked [total,5] observation afterwards."""
        obs, rew, done = self.backend.step(local_actions)
        if self.gather_obs:
            o = torch.as_tensor(np.ascontiguousarray(obs)) if not torch.is_tensor(obs) else obs
            o = o.to(torch.float32).contiguous()
            if self._gathered is None:
                self._gathered = torch.empty(self.total, o.shape[1], dtype=torch.float32, device=o.device)
            dist.all_gather_into_tensor(self._gathered, o)
        return obs, rew, done

    @property
    def obs_all(self):
        return self._gathered
