"""Agent — the rollout buffer + n-step A3C/GAE loss of player_util.py:9-161, batched over N envs.

Same constructor, method and attribute names as the reference so train/test/eval-style callers read the same:
    Agent(model, env, args, state, device); action_train(); action_test(); reset(); optimize(params, optimizer,
    shared_model, training_mode, device_share); .done .reward .reward_org .eps_len .n_steps .num_agents ...

What changes under vectorisation (SURVEY.md §7 hard part 5):
  * env is a VecEnv (environment.py): N envs advance per call; finished envs auto-reset inside the step launch,
    so the rollout never breaks early — per-env episode boundaries are carried as `done` masks instead:
      - LSTM state of a finished env is zeroed before its next step (reset_rnn_hiden, player_util.py:98-102);
      - n-step return and GAE are cut at the boundary (R = r, no bootstrap — the `done` branch of :109-117);
  * the loss is the mean over envs of the reference's per-env loss (:128-154);
  * ensure_shared_grads + lock-free SharedAdam (utils.py:36-44) become an all-reduce(mean) of the flat gradient
    bucket over RCCL followed by an identical optimizer step on every replica.
Reference quirk kept on purpose: clip_grad_norm_(params, 50) at :157 is a no-op in the reference (the generator
is exhausted after the first call, which itself sees grad=None), so no clipping is applied
unless args.max_grad_norm is set.
"""
import torch
import torch.distributed as dist


class Agent(object):
    def __init__(self, model, env, args, state, device):
        self.model = model
        self.env = env
        self.num_agents = len(env.observation_space)
        self.num_envs = getattr(env, "num_envs", 1)
        if 'continuous' in args.network:
            raise NotImplementedError("continuous actions belong to the Unreal envs (out of scope)")
        self.dim_action = 1
        self.args = args
        self.device = device
        self.rnn_out = args.rnn_out
        self.values, self.log_probs, self.rewards, self.entropies, self.preds, self.dones = [], [], [], [], [], []
        self.states, self.actions, self.h0, self.c0 = [], [], None, None
        self._buf, self._pending_done, self._cache, self._actions_buf = None, None, None, None
        self.cache_rollout = True   # fast path on the GPU: learner back-propagates through the actor's forward pass
        self.fused_heads = True     # ... and evaluates heads + loss terms as one fused HIP node per player
        self.fused_bookkeeping = True   # rollout prologue / epilogue as one launch each (csrc/driver_hip.hip)
        self._keep = None
        self.carry_out, self.carry_written, self._want_loss_terms = None, (), False
        self._one = torch.ones((), dtype=torch.float32, device=device)   # seed of the backward pass (made outside any capture)
        self.done = torch.ones(self.num_envs, dtype=torch.uint8, device=device)
        self.info = None
        self.reward = 0
        self.reward_org = 0
        self.num_steps = 0
        self.n_steps = 0
        self.state = state
        self.eps_len = torch.zeros(self.num_envs, dtype=torch.int32, device=device)
        self.hxs = torch.zeros(self.num_envs, self.num_agents, self.rnn_out, device=device)
        self.cxs = torch.zeros(self.num_envs, self.num_agents, self.rnn_out, device=device)
        self.w_entropy_target = getattr(args, "entropy_target", 0.2)
        self.gpu_id = device.index if isinstance(device, torch.device) and device.index is not None else -1

    # -- acting ------------------------------------------------------------------------------------------
    def action_train(self):
        self.n_steps += 1
        value_multi, action_env_multi, entropy, log_prob, (self.hxs, self.cxs), R_pred = self.model(
            (self.state, (self.hxs, self.cxs)))
        state_multi, reward_multi, done, self.info = self.env.step(action_env_multi)
        self.reward_org = reward_multi
        self.reward = reward_multi
        self.state = state_multi
        self.done = done
        keep = (done == 0)
        self.eps_len = (self.eps_len + 1) * keep.to(self.eps_len.dtype)
        # a finished env starts its next episode with a zero hidden state (train.py:73-74 -> reset())
        k = keep.to(self.hxs.dtype).view(-1, 1, 1)
        self.hxs = self.hxs * k
        self.cxs = self.cxs * k
        self.values.append(value_multi)
        self.entropies.append(entropy)
        self.log_probs.append(log_prob)
        self.rewards.append(reward_multi.unsqueeze(2))
        self.preds.append(R_pred)
        self.dones.append(done)
        return self

    def action_rollout(self):
        """Actor half of the fast path: the same step as action_train, but only what acting needs is computed
        (model.act, no autograd graph); what the learner needs to re-evaluate the step (state, actions, done) is
        stored instead. LSTM states are kept per player as contiguous [N,R] tensors during the rollout."""
        self.n_steps += 1
        stepped = False
        if self._cache is not None:
            # where the kernels allow it the env step runs inside the policy step's last launch (k_act_step)
            env_out = None
            if self._buf is not None and hasattr(self.env, "fused_step_out"):
                t = len(self.states)
                env_out = self.env.fused_step_out((self._buf[0][t + 1], self._buf[1][t], self._buf[2][t]))
            actions = self.model.act_cached(self.state, self._cache, len(self.states), self._pending_done, env_out=env_out)
            stepped = env_out is not None and getattr(self.model, "env_stepped", False)
        elif hasattr(self.model, "act") and self.num_agents == 2 and not getattr(self.model, "single", False):
            actions, self._hs, self._cs = self.model.act(self.state, self._hs, self._cs, self._pending_done)
        else:
            self._apply_pending_done()
            with torch.no_grad():
                _, actions, _, _, (hx, cx), _ = self.model((self.state, (torch.stack(self._hs, 1), torch.stack(self._cs, 1))))
            self._hs, self._cs = list(hx.unbind(1)), list(cx.unbind(1))
        self.states.append(self.state)
        if self._actions_buf is None:                            # else the sampler wrote them in place
            self.actions.append(torch.stack(actions, 1))
        if stepped:                                     # the step's outputs are already in the rollout storage
            state_multi, reward_multi, done, self.info = self.env.after_fused_step(env_out)
        elif self._buf is not None:                     # the step kernel writes straight into the rollout storage
            t = len(self.states) - 1
            state_multi, reward_multi, done, self.info = self.env.step(
                actions, out=(self._buf[0][t + 1], self._buf[1][t], self._buf[2][t]))
        else:
            state_multi, reward_multi, done, self.info = self.env.step(actions)
        self.reward_org = reward_multi
        self.reward = reward_multi
        self.state = state_multi
        self.done = done
        self._pending_done = done                                # finished envs restart from a zero LSTM state:
        self.rewards.append(reward_multi.unsqueeze(2))           # applied by the next act() / end_rollout
        self.dones.append(done)
        return self

    def _apply_pending_done(self):
        if self._cache is not None:                               # state of the last step lives in the cache
            T = len(self.states)
            self._hs = list(self._cache.h_all[:, T].unbind(0))
            self._cs = list(self._cache.c_all[:, T].unbind(0))
        if self._pending_done is not None:
            k = (self._pending_done == 0).to(self._hs[0].dtype).unsqueeze(1)
            self._hs = [h * k for h in self._hs]
            self._cs = [c * k for c in self._cs]
            self._pending_done = None

    def begin_rollout(self, num_steps=None):
        """Remember the LSTM state the rollout starts from (the learner re-runs the recurrence from it). With
        num_steps and an env that offers rollout_buffers, the rollout is stored in place (no stacking copies)."""
        self._buf = None
        if num_steps is not None and hasattr(self.env, "rollout_buffers"):
            self._buf = self.env.rollout_buffers(num_steps)
        self.update_rnn_hiden()
        self.h0, self.c0 = self.hxs, self.cxs
        self.states, self.actions = [], []
        self._pending_done = None
        self._cache = None
        self._keep = None
        sampler = getattr(self.model, "_sampler", None)
        if (sampler is None and getattr(self.model, "fused_sampling", False) and hasattr(self.model, "new_cache")
                and torch.is_tensor(self.state) and self.state.is_cuda):
            # made here and not lazily inside the first step: without an open draw block the first rollout of a model would
            # take the slower per-player launches instead of the path every later rollout takes
            from . import fused
            sampler = self.model._sampler = fused.ActionSampler(self.state.device)
        obs0 = self.state.reshape(self._buf[0][0].shape) if self._buf is not None else None
        want_cache = num_steps is not None and self.cache_rollout and hasattr(self.model, "new_cache") and self.num_agents == 2
        # the rollout's first launch (atr_rollout_begin2) moves the LSTM state and the observation into the stores AND makes the
        # per-rollout constants of the actor + the draw counter's bump: one launch instead of ~8
        one_launch = (want_cache and self.fused_bookkeeping and torch.is_tensor(self.hxs) and self.hxs.is_cuda
                      and self.hxs.is_contiguous() and self.cxs.is_contiguous() and sampler is not None
                      and (obs0 is None or (obs0.dtype == self._buf[0].dtype and obs0.is_contiguous()
                                            and (obs0.numel() * obs0.element_size()) % 4 == 0)))
        if want_cache:
            # will every step's env.step run inside the policy step's last launch? (action_rollout asks the same question per
            # step; envs that decline — RPF targets, 'Full' observations, --rescale, stacked frames, NumpyVecEnv — step on
            # their own, and the cache must then not take the form whose hidden rows only k_act_step writes)
            env_fused = (self._buf is not None and hasattr(self.env, "fused_step_out") and num_steps >= 1
                         and self.env.fused_step_out((self._buf[0][1], self._buf[1][0], self._buf[2][0])) is not None)
            self._cache = self.model.new_cache(num_steps, self.state, defer_consts=one_launch, env_fused=env_fused)
        self._actions_buf = getattr(self._cache, "actions", None)
        consts = getattr(self._cache, "consts", None)
        if consts is not None:
            from . import fused
            if not (one_launch and fused.rollout_consts_ok(consts)):
                self.model.fill_consts(self._cache)               # (tensor ops; the launch below then only moves state)
                consts = None
        if sampler is not None:
            sampler.begin_block(launch=consts is None)            # one counter bump per rollout, ordinals inside
        if self._cache is not None:                               # LSTM state lives in the cache: slot t -> t+1
            if one_launch:
                from . import fused                               # both copies + the observation's in one launch
                fused.rollout_begin(self.hxs, self.cxs, self._cache.h_all, self._cache.c_all, obs0,
                                    self._buf[0][0] if obs0 is not None else None, consts=consts,
                                    counter=sampler.counter if consts is not None else None)
                if consts is None:
                    self._seed_fh()
                else:
                    self._cache.consts = None
                return
            self._cache.h_all[:, 0].copy_(self.hxs.transpose(0, 1))
            self._cache.c_all[:, 0].copy_(self.cxs.transpose(0, 1))
            self._seed_fh()
            if obs0 is not None:
                self._buf[0][0].copy_(obs0)
            return
        if obs0 is not None:
            self._buf[0][0].copy_(obs0)
        self._hs = [h.contiguous() for h in self.hxs.unbind(1)]
        self._cs = [c.contiguous() for c in self.cxs.unbind(1)]
        if hasattr(self.model, "begin_act"):
            self.model.begin_act()

    def _seed_fh(self):
        """The h columns of slot 0 of the cache's [features | k h_prev] rows (the one-GEMM LSTMCell path of model._act_step):
        the state the rollout starts from — masked by the previous rollout's last done flags already (end_rollout)."""
        fh = getattr(self._cache, "fh_all", None)
        if fh is not None:
            R = self._cache.h_all.shape[-1]
            fh[:, 0, :, fh.shape[-1] - R:].copy_(self._cache.h_all[:, 0])

    def end_rollout(self):
        """Publish the per-player LSTM states back as hxs/cxs [N,A,R] and the episode-length counters."""
        T = len(self.states)
        if (self._cache is not None and self._buf is not None and self.fused_bookkeeping and T == self._cache.T
                and T == self._buf[2].shape[0] and self._cache.h_all.is_cuda and self.eps_len.dtype == torch.int32):
            from . import fused   # final state masked by the last done, episode lengths, the keep mask: one launch
            if getattr(self.model, "_sampler", None) is not None:
                self.model._sampler.end_block()
            if hasattr(self.model, "_bsum"):
                self.model._bsum = None
            N, R = self.num_envs, self._cache.h_all.shape[-1]
            self.eps_len = self.eps_len.contiguous()
            self._keep = torch.empty((T, N), device=self.device)
            # carry_out (set by the graphed drivers): the tensors the NEXT rollout starts from. The final LSTM state goes
            # straight there (this rollout's begin already copied the old one into the store), and the same launch publishes
            # the last observation and done flags — instead of four copy launches after the learner
            co = getattr(self, "carry_out", None)
            self.carry_written = ()
            if (co is not None and co["hxs"].shape == (N, 2, R) and co["hxs"].is_contiguous() and co["cxs"].is_contiguous()
                    and co["state"].dtype == self.state.dtype and co["state"].numel() == self.state.numel()
                    and self.state.is_contiguous() and co["state"].is_contiguous()
                    and (self.state.numel() * self.state.element_size()) % 4 == 0
                    and co["done"].dtype == torch.uint8 and co["done"].is_contiguous()):
                self.hxs, self.cxs = co["hxs"], co["cxs"]
                fused.rollout_end(self._cache.h_all, self._cache.c_all, self._buf[2], self.hxs, self.cxs, self.eps_len,
                                  self._keep, obs_src=self.state, obs_dst=co["state"], done_dst=co["done"])
                self.carry_written = ("state", "hxs", "cxs", "done")
            else:
                self.hxs = torch.empty((N, 2, R), device=self.device)
                self.cxs = torch.empty((N, 2, R), device=self.device)
                fused.rollout_end(self._cache.h_all, self._cache.c_all, self._buf[2], self.hxs, self.cxs, self.eps_len,
                                  self._keep)
            self._pending_done = None
            return
        self._apply_pending_done()
        if getattr(self.model, "_sampler", None) is not None:
            self.model._sampler.end_block()
        if hasattr(self.model, "_bsum"):
            self.model._bsum = None                               # per-rollout cache; the weights change next
        self.hxs, self.cxs = torch.stack(self._hs, 1), torch.stack(self._cs, 1)
        dones = self._buf[2] if self._buf is not None and len(self.dones) == self._buf[2].shape[0] \
            else torch.stack(self.dones, 0)
        nd = (dones == 0).to(self.eps_len.dtype)                              # [T, N]
        alive_since = torch.flip(torch.cumprod(torch.flip(nd, [0]), 0), [0])  # 1 while no done from t to the end
        self.eps_len = self.eps_len * alive_since[0] + alive_since.sum(0)

    def action_test(self):
        with torch.no_grad():
            value_multi, action_env_multi, entropy, log_prob, (self.hxs, self.cxs), R_pred = self.model(
                (self.state, (self.hxs, self.cxs)), True)
        state_multi, self.reward, done, self.info = self.env.step(action_env_multi)
        self.state = state_multi
        self.done = done
        keep = (done == 0)
        self.eps_len = (self.eps_len + 1) * keep.to(self.eps_len.dtype)
        k = keep.to(self.hxs.dtype).view(-1, 1, 1)
        self.hxs = self.hxs * k
        self.cxs = self.cxs * k
        return self

    def reset(self):
        self.state = self.env.reset()
        self.num_agents = self.state.shape[1]
        self.eps_len = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self.done = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
        self.reset_rnn_hiden()

    def clear_actions(self):
        self.values, self.log_probs, self.rewards, self.entropies, self.preds, self.dones = [], [], [], [], [], []
        self.states, self.actions = [], []
        return self

    def reset_rnn_hiden(self):
        self.cxs = torch.zeros(self.num_envs, self.num_agents, self.rnn_out, device=self.device)
        self.hxs = torch.zeros(self.num_envs, self.num_agents, self.rnn_out, device=self.device)

    def update_rnn_hiden(self):
        self.cxs = self.cxs.detach()
        self.hxs = self.hxs.detach()

    # -- learning ----------------------------------------------------------------------------------------
    def loss(self, training_mode):
        """The loss of player_util.py:108-154 per env, averaged over envs. Returns (loss, policy_loss [N,A,1],
        value_loss [N,A,1], entropies [N,A,1], pred_loss [N,1])."""
        args = self.args
        N, A = self.num_envs, self.num_agents
        with torch.no_grad():  # bootstrap value V(s_T) (:109-117); finished envs are masked step by step below
            value_multi, _, _, _, _, _ = self.model((self.state, (self.hxs, self.cxs)))
        R = value_multi.detach()
        self.values.append(R)
        policy_loss = torch.zeros(N, A, 1, device=self.device)
        value_loss = torch.zeros(N, A, 1, device=self.device)
        pred_loss = torch.zeros(N, 1, device=self.device)
        entropies = torch.zeros(N, A, self.dim_action, device=self.device)
        w_entropies = float(args.entropy) * torch.ones(1, A, self.dim_action, device=self.device)
        if A > 1:
            w_entropies[:, 1:] = float(self.w_entropy_target)
        gae = torch.zeros(N, A, 1, device=self.device)
        use_aux = 'reward' in args.aux
        for i in reversed(range(len(self.rewards))):
            nd = (self.dones[i] == 0).to(R.dtype).view(N, 1, 1)
            if use_aux:
                pred_loss = pred_loss + (self.preds[i] - self.rewards[i][:, 0]).abs()   # L1Loss (:129-130)
            R = args.gamma * R * nd + self.rewards[i]
            advantage = R - self.values[i]
            value_loss = value_loss + 0.5 * advantage.pow(2)
            delta_t = self.rewards[i] + args.gamma * self.values[i + 1].detach() * nd - self.values[i].detach()
            gae = gae * args.gamma * args.tau * nd + delta_t
            policy_loss = policy_loss - (self.log_probs[i] * gae.detach()) - (w_entropies * self.entropies[i])
            entropies = entropies + self.entropies[i].detach()
        loss_tracker = (policy_loss[:, 0] + 0.5 * value_loss[:, 0]).mean()
        loss_target = (policy_loss[:, 1] + 0.5 * value_loss[:, 1]).mean() if A > 1 else 0
        if training_mode == 0:
            loss = loss_tracker
        elif training_mode == 1:
            loss = loss_target
        else:
            loss = loss_tracker + loss_target
        if use_aux and training_mode != 0:
            loss = loss + pred_loss.mean()
        return loss, policy_loss, value_loss, entropies, pred_loss

    def loss_recompute(self, training_mode):
        """Learner half of the fast path: the loss of player_util.py:108-154 over the stored rollout, evaluated
        time-batched. The n-step returns R_t and the GAE terms use only rewards, done flags and DETACHED values
        (`.data` at :135), so they are constants computed first; the differentiable part is then one vectorised
        expression over [T, N, A]. Same value and gradients as loss() on an action_train rollout."""
        args = self.args
        N, A, T = self.num_envs, self.num_agents, len(self.rewards)
        dev = self.device
        if self._actions_buf is not None:
            actions = self._actions_buf.transpose(1, 2)                      # [T, N, A] view of the [T, A, N] store
        else:
            actions = torch.stack(self.actions, 0)
        if self._buf is not None and T == self._buf[1].shape[0]:
            states = self._buf[0][:T].unsqueeze(3).unsqueeze(4)              # [T, N, A, 1, 1, h, w] views
            rewards = self._buf[1].unsqueeze(3)
            nd = self._keep if self._keep is not None else (self._buf[2] == 0).to(rewards.dtype)
        else:
            states = torch.stack(self.states, 0)
            rewards = torch.stack(self.rewards, 0)                           # [T, N, A, 1]
            nd = (torch.stack(self.dones, 0) == 0).to(rewards.dtype)         # [T, N]
        if self._cache is not None and T == self._cache.T and self.fused_heads:
            return self._loss_fused_heads(training_mode, states, actions, rewards, nd)
        if self._cache is not None and T == self._cache.T:
            values, entropies, log_probs, preds = self.model.forward_sequence_cached(self._cache, states, actions, nd)
        else:
            values, entropies, log_probs, preds = self.model.forward_sequence(states, actions, self.h0, self.c0, nd)
        with torch.no_grad():
            boot, _, _, _, _, _ = self.model((self.state, (self.hxs, self.cxs)))
            v = torch.cat([values.detach(), boot.unsqueeze(0)], 0)           # [T+1, N, A, 1]
            ndv = nd.view(T, N, 1, 1)
            if v.is_cuda and v.dtype == torch.float32:
                from . import fused
                R, gae = fused.gae_returns(rewards, v, nd, args.gamma, args.tau)   # one launch (csrc/lstm_hip.hip)
            else:
                R = torch.empty_like(rewards)
                gae = torch.empty_like(rewards)
                r_run, g_run = v[T], torch.zeros_like(v[T])
                for i in reversed(range(T)):
                    r_run = args.gamma * r_run * ndv[i] + rewards[i]
                    delta_t = rewards[i] + args.gamma * v[i + 1] * ndv[i] - v[i]
                    g_run = g_run * args.gamma * args.tau * ndv[i] + delta_t
                    R[i], gae[i] = r_run, g_run
        w_entropies = float(args.entropy) * torch.ones(1, 1, A, 1, device=dev)
        if A > 1:
            w_entropies[:, :, 1:] = float(self.w_entropy_target)
        value_loss = (0.5 * (R - values).pow(2)).sum(0)                      # [N, A, 1]
        policy_loss = (-(log_probs * gae) - w_entropies * entropies).sum(0)
        use_aux = 'reward' in args.aux
        pred_loss = (preds - rewards[:, :, 0]).abs().sum(0) if use_aux else torch.zeros(N, 1, device=dev)
        loss_tracker = (policy_loss[:, 0] + 0.5 * value_loss[:, 0]).mean()
        loss_target = (policy_loss[:, 1] + 0.5 * value_loss[:, 1]).mean() if A > 1 else 0
        if training_mode == 0:
            loss = loss_tracker
        elif training_mode == 1:
            loss = loss_target
        else:
            loss = loss_tracker + loss_target
        if use_aux and training_mode != 0:
            loss = loss + pred_loss.mean()
        return loss, policy_loss, value_loss, entropies.detach().sum(0), pred_loss

    def _loss_fused_heads(self, training_mode, states, actions, rewards, nd):
        """loss_recompute over a cached rollout with each player's heads + loss terms as one fused HIP node
        (fused.heads_loss, csrc/heads_hip.hip): values first (the returns / GAE need them detached), then per player
        one launch that yields the objective contribution, dL/dh and the head gradients. Returns the same tuple as
        loss_recompute, the per-env statistics already averaged over envs (shape [1, A, 1] / [1, 1])."""
        from . import fused
        args, model = self.args, self.model
        T, N, A = self._cache.T, self.num_envs, self.num_agents
        dev = self.device
        players = (model.player0, model.player1)
        # (train-mode 0 / 1: the other player's loss terms carry coefficient 0 — its recurrence is not back-propagated)
        need = (training_mode != 1, training_mode != 0) if A == 2 else None
        h_seq = model.cached_hidden(self._cache, states, actions, nd, need)   # per player [T, N, R], with history
        R_dim = h_seq[0].shape[-1]
        v = torch.empty((T + 1, N, A, 1), dtype=torch.float32, device=dev)
        with torch.no_grad():
            if A == 2:
                fused.heads_values2([h_seq[p].detach().reshape(T * N, R_dim) for p in range(2)],
                                    [players[p].critic.critic_linear for p in range(2)], v)
            else:
                for p in range(A):
                    fused.heads_values(h_seq[p].detach().reshape(T * N, R_dim), players[p].critic.critic_linear, v, p)
            sampler = getattr(model, "_sampler", None)
            if hasattr(model, "boot_values") and sampler is not None and getattr(sampler, "_last", None) is not None:
                sampler.reopen_block()       # V(s_T) with the rollout's kernels: one more actor step + the critic heads
                model.boot_values(self.state, self._cache, self.done, v[T])
                sampler.end_block()
            else:
                # (the LSTM state the rollout ended on, from THIS rollout's own store: under the graphed drivers self.hxs /
                # self.cxs alias the carry, which the next rollout may already be overwriting on another stream)
                k_end = (self.done == 0).to(torch.float32).view(N, 1, 1)
                hx_end = self._cache.h_all[:, T].transpose(0, 1) * k_end
                cx_end = self._cache.c_all[:, T].transpose(0, 1) * k_end
                boot, _, _, _, _, _ = model((self.state, (hx_end, cx_end)))
                v[T].copy_(boot)
            R, gae = fused.gae_returns(rewards.contiguous(), v, nd, args.gamma, args.tau)
        use_aux = 'reward' in args.aux and getattr(model, "tat", False) and getattr(model.player1, "sub_task", False)
        w_ent = [float(args.entropy)] + [float(self.w_entropy_target)] * (A - 1)
        scale = [1.0 / N if training_mode in (0, -1) else 0.0, 1.0 / N if training_mode in (1, -1) else 0.0]
        if training_mode not in (0, 1, -1):
            scale = [1.0 / N, 1.0 / N]
        scale_aux = 1.0 / N if (use_aux and training_mode != 0) else 0.0
        rew_c = rewards.contiguous()
        if A == 2 and players[0].actor.actor_linear.weight.shape == players[1].actor.actor_linear.weight.shape:
            # both players' heads + loss terms: one launch + one reduction launch, actions read in place from the rollout's
            # [T, players, N] store, statistics already averaged over envs
            cfg = []
            for p in range(2):
                aux = players[p].reward_aux if (p == 1 and use_aux) else None
                cfg.append(dict(actions=actions[:, :, p], ret=R, gae=gae, val=v, off=p, r_aux=rew_c if aux is not None else None,
                                aux_off=0, scale=scale[p], scale_aux=scale_aux if aux is not None else 0.0, w_ent=w_ent[p],
                                stats_scale=1.0 / N))
            auxes = [None, players[1].reward_aux if use_aux else None]
            l0, l1, st = fused.heads_loss_pair([h_seq[p].reshape(T * N, R_dim) for p in range(2)],
                                               [players[p].actor.actor_linear for p in range(2)],
                                               [players[p].critic.critic_linear for p in range(2)], auxes, cfg)
            # the objective is l0 + l1; compute_grads differentiates the two terms directly (no add launches, no seed fills)
            loss = (l0, l1) if getattr(self, "_want_loss_terms", False) else l0 + l1
        else:
            loss, stats = 0, []
            for p in range(A):
                aux = players[p].reward_aux if (p == 1 and use_aux) else None
                lp, st_ = fused.heads_loss(h_seq[p].reshape(T * N, R_dim), players[p].actor.actor_linear,
                                           players[p].critic.critic_linear, aux, actions[:, :, p].reshape(T * N), R, gae, v, p,
                                           rew_c if aux is not None else None, 0, scale[p], scale_aux if aux is not None else 0.0,
                                           w_ent[p], unit_coeff=True)   # summed and differentiated as is
                loss = loss + lp
                stats.append(st_)
            st = torch.stack(stats, 0) / N                                   # [A, 4]: policy, value, entropy, |aux|
        policy_loss, value_loss, entropies = (st[:, k].reshape(1, A, 1) for k in range(3))
        pred_loss = st[1, 3].reshape(1, 1) if (A > 1 and use_aux) else torch.zeros(1, 1, device=dev)
        return loss, policy_loss, value_loss, entropies, pred_loss

    def compute_grads(self, optimizer, training_mode):
        """loss -> backward into the flat gradient bucket (hipGraph-capturable: no host sync)."""
        fast = len(self.states) > 0
        bucket = getattr(optimizer, "bucket", None)
        self._want_loss_terms = bucket is not None and hasattr(bucket, "set_grads")
        try:
            loss, policy_loss, value_loss, entropies, pred_loss = (self.loss_recompute if fast else self.loss)(training_mode)
        finally:
            self._want_loss_terms = False
        if bucket is not None and hasattr(bucket, "set_grads"):
            from . import fused
            terms = list(loss) if isinstance(loss, (tuple, list)) else [loss]
            if self._one is None or self._one.device != terms[0].device:     # (CPU agents; GPU ones made it in __init__)
                self._one = torch.ones((), dtype=terms[0].dtype, device=terms[0].device)
            # the weight-gradient GEMMs of the pass register themselves and go out as one grouped launch at the end
            with fused.deferred_weight_grads(bucket) as q:
                grads = torch.autograd.grad(terms, bucket.params, grad_outputs=[self._one] * len(terms), allow_unused=True)
                if q is not None:
                    q.check(grads)
                    q.flush()
            bucket.set_grads(grads)
        else:
            optimizer.zero_grad()
            loss.backward()
        self.clear_actions()
        if hasattr(self.model, "cache_dense"):
            self.model.cache_dense(False)
        def env_mean(x, keepdim=False):      # the fused loss already reports means over envs (leading dim 1): no launch
            x = x.detach()
            if x.shape[0] == 1:
                return x if keepdim else x[0]
            return x.mean(0, keepdim=keepdim)
        return env_mean(policy_loss), env_mean(value_loss), env_mean(entropies), env_mean(pred_loss, keepdim=True)

    def allreduce_grads(self, optimizer):
        """The ONE collective of the path: flat fp32 gradient bucket, mean over ranks (RCCL over xGMI). Replaces
        ensure_shared_grads + the shared-memory model (utils.py:36-44)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        # (a 1-rank group issues nothing unless ATR_FORCE_ALLREDUCE=1: tools/multirank_probe.py measures the collective's
        # stream hand-overs on the one leased GPU that way)
        if dist.get_world_size() == 1 and __import__("os").environ.get("ATR_FORCE_ALLREDUCE") != "1":
            return
        world = dist.get_world_size()
        bucket = getattr(optimizer, "bucket", None)
        if bucket is not None:
            if bucket.grad.is_cuda and dist.get_backend() == "nccl":
                # RCCL averages inside the collective (ncclAvg): one call, no separate divide launch between the two graphs
                dist.all_reduce(bucket.grad, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(bucket.grad, op=dist.ReduceOp.SUM)
                bucket.grad.div_(world)
        else:
            for p in self.model.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                    p.grad.div_(world)

    def optimize(self, params, optimizer, shared_model, training_mode, device_share):
        """One synchronous data-parallel update. `params`, `shared_model`, `device_share` are accepted for
        signature compatibility with player_util.py:108; the replica IS the shared model."""
        stats = self.compute_grads(optimizer, training_mode)
        self.allreduce_grads(optimizer)
        max_norm = getattr(self.args, "max_grad_norm", None)
        if max_norm:
            if getattr(optimizer, "bucket", None) is not None:
                from .train import clip_flat_grad_           # the same expression the graphed drivers capture
                clip_flat_grad_(optimizer, max_norm)
            else:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm)
        optimizer.step()
        return stats
