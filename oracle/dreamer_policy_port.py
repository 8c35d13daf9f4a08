"""TEST INFRASTRUCTURE (oracle/): NumPy restatement of the reference's DEPLOYED Dreamer agent - the trained policy its ROS node runs.

What it follows (reference files, read as text; nothing is imported from them - TensorFlow is not in this image):
  * ros_agent/models/dreamer/racing_dreamer.py:9-80   RacingDreamer: RSSM(stoch 30, deter 200, hidden 200, ELU), ActionDecoder(size 2,
    4 layers x 400 units, 'tanh_normal', init_std 5.0), `_preprocess_lidar` (clip to [0, 15] m, / 15 - 0.5; the embedding IS the scan:
    encoder.pkl is empty), `action` (obs_step from the previous latent and the previous RAW action, feature = [stoch, deter], actor mode),
    `postprocess_action` (clip to +-1, map to the reduced space [0.005, 1] x [-1, 1] = ReduceActionSpace, dreamer/wrappers.py:128-130).
  * ros_agent/models/dreamer/models.py:61-87            RSSM.obs_step / img_step (Dense img1 -> GRUCell -> deter; Dense obs1, obs2 on
    [deter, embed] -> mean, softplus(std) + 0.1; the posterior is SAMPLED).
  * ros_agent/models/dreamer/models.py:339-353          ActionDecoder 'tanh_normal': mean = 5 tanh(mean / 5), std = softplus(std + raw_init_std) + 1e-4,
    tanh-transformed normal wrapped in SampleDist.
  * dreamer/tools.py:318-321                            SampleDist.mode: the one of 100 samples with the highest log-probability.
  * tf.keras.layers.GRUCell (TF 2 defaults: reset_after=True - the checkpoint's (2, 600) bias says so -, sigmoid gates, tanh candidate).
The weights are the reference's own checkpoint files (ros_agent/checkpoints/<name>/{rssm,actor,reward}.pkl: tuples of float32 arrays in
`tf.Module.variables` order), converted by tests/golden/make_golden_dreamer_policy.py, which unpickles them with an allow-list
(numpy array reconstruction only).

Two modes: `sample=True` draws what the reference draws (posterior sample, best-of-100 action) from a seeded NumPy generator -
statistically the reference's behaviour, not its random stream; `sample=False` uses the posterior mean and tanh(mean): deterministic.
"""
import numpy as np

f32 = np.float32
RAW_INIT_STD = float(np.log(np.exp(5.0) - 1.0))
RSSM_KEYS = ("gru_kernel", "gru_recurrent", "gru_bias", "img1_w", "img1_b", "img2_w", "img2_b", "img3_w", "img3_b",
             "obs1_w", "obs1_b", "obs2_w", "obs2_b")
ACTOR_KEYS = ("h0_w", "h0_b", "h1_w", "h1_b", "h2_w", "h2_b", "h3_w", "h3_b", "hout_w", "hout_b")
REWARD_KEYS = ("reward_h0_w", "reward_h0_b", "reward_h1_w", "reward_h1_b", "reward_hout_w", "reward_hout_b")
# actor_version "normalized" (racing_dreamer.py:23-25; models.py:354-364): batch normalisation of the output layer, no mean scaling.
# Order inside the checkpoint as identified by the moments of the layer's input on live states: moving mean, moving variance, gamma, beta.
ACTOR_NORM_KEYS = ACTOR_KEYS[:8] + ("hnorm_mean", "hnorm_var", "hnorm_gamma", "hnorm_beta") + ACTOR_KEYS[8:]
# LidarOccupancyDecoder (dreamer/models.py:444-465): Dense 230 -> 64, then four stride-2 'valid' Conv2DTranspose layers with ReLU
# (1 -> 5 -> 13 -> 30 -> 64 pixels; kernels [kh, kw, out, in]); the output are the Bernoulli logits of the 64 x 64 lidar_occupancy image
DECODER_KEYS = ("dec_h1_w", "dec_h1_b", "dec_h2_k", "dec_h2_b", "dec_h3_k", "dec_h3_b", "dec_h4_k", "dec_h4_b", "dec_h5_k", "dec_h5_b")


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(f32)


def softplus(x):
    return (np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))).astype(f32)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-np.clip(x, -60.0, 60.0)))).astype(f32)


class DreamerPolicy:
    def __init__(self, weights, sample=True, seed=0):
        """weights: mapping with RSSM_KEYS and ACTOR_KEYS (float32 arrays)."""
        self.w = {k: np.asarray(weights[k], f32) for k in RSSM_KEYS + ACTOR_NORM_KEYS + REWARD_KEYS + DECODER_KEYS if k in weights}
        self.normalized = "hnorm_gamma" in self.w
        assert self.w["obs1_w"].shape == (200 + 1080, 200) and self.w["h0_w"].shape == (230, 400)
        self.sample = bool(sample)
        self.rng = np.random.default_rng(seed)

    def initial(self, n):
        return dict(stoch=np.zeros((n, 30), f32), deter=np.zeros((n, 200), f32), action=np.zeros((n, 2), f32))

    @staticmethod
    def preprocess(scan_m):
        return (np.clip(np.asarray(scan_m, f32), 0.0, 15.0) / f32(15.0) - f32(0.5)).astype(f32)

    def predicted_reward(self, state):
        """The reference's reward head on the feature of `state` (models.py:301-318: two ELU layers of 400, mean of a unit
        normal): what the world model, trained on the reference simulator's rewards, expects for the step that led here."""
        w = self.w
        h = np.concatenate([state["stoch"], state["deter"]], 1)
        for i in range(2):
            h = elu(h @ w[f"reward_h{i}_w"] + w[f"reward_h{i}_b"])
        return (h @ w["reward_hout_w"] + w["reward_hout_b"])[:, 0].astype(f32)

    def decoded_occupancy(self, state):
        """Bernoulli logits [n, 64, 64] of the lidar_occupancy image the reference's decoder reconstructs from `state`."""
        w = self.w
        x = (np.concatenate([state["stoch"], state["deter"]], 1) @ w["dec_h1_w"] + w["dec_h1_b"]).reshape(-1, 1, 1, 64)
        for name in ("dec_h2", "dec_h3", "dec_h4", "dec_h5"):
            k, b = w[name + "_k"], w[name + "_b"]
            n, hh, ww, _ = x.shape
            out = np.zeros((n, (hh - 1) * 2 + k.shape[0], (ww - 1) * 2 + k.shape[1], k.shape[2]), f32)
            for u in range(k.shape[0]):
                for v in range(k.shape[1]):
                    out[:, u:u + 2 * hh:2, v:v + 2 * ww:2, :] += np.einsum("bhwc,oc->bhwo", x, k[u, v])
            x = np.maximum(out + b, 0.0)
        return x[..., 0]

    def _gru(self, x, h):
        w = self.w
        mx = x @ w["gru_kernel"] + w["gru_bias"][0]
        mh = h @ w["gru_recurrent"] + w["gru_bias"][1]
        xz, xr, xh = np.split(mx, 3, axis=1)
        hz, hr, hh = np.split(mh, 3, axis=1)
        z, r = sigmoid(xz + hz), sigmoid(xr + hr)
        cand = np.tanh(xh + r * hh)
        return (z * h + (1.0 - z) * cand).astype(f32)

    def act(self, scan_m, state, reset=None):
        """scan_m float32 [n, 1080] in metres (the env's beam order); state from `initial` / the previous call; reset: bool [n],
        envs whose episode has just begun (their latent and previous action start from zero, as a fresh `state=None` does).
        Returns (raw action float32 [n, 2] in [-1, 1] = (motor, steering) BEFORE ReduceActionSpace, new state)."""
        w = self.w
        n = len(scan_m)
        stoch, deter, prev = state["stoch"], state["deter"], state["action"]
        if reset is not None and np.any(reset):
            keep = (~np.asarray(reset, bool))[:, None].astype(f32)
            stoch, deter, prev = stoch * keep, deter * keep, prev * keep
        embed = self.preprocess(scan_m)
        x = elu(np.concatenate([stoch, prev], 1) @ w["img1_w"] + w["img1_b"])
        deter = self._gru(x, deter)
        x = elu(np.concatenate([deter, embed], 1) @ w["obs1_w"] + w["obs1_b"])
        x = x @ w["obs2_w"] + w["obs2_b"]
        mean, std = x[:, :30], softplus(x[:, 30:]) + f32(0.1)
        stoch = (mean + std * self.rng.standard_normal(mean.shape).astype(f32)).astype(f32) if self.sample else mean.astype(f32)
        h = np.concatenate([stoch, deter], 1)
        for i in range(4):
            h = elu(h @ w[f"h{i}_w"] + w[f"h{i}_b"])
        out = h @ w["hout_w"] + w["hout_b"]
        if self.normalized:             # inference-mode batch normalisation (Keras epsilon 1e-3), linear mean
            out = (out - w["hnorm_mean"]) / np.sqrt(w["hnorm_var"] + f32(1e-3)) * w["hnorm_gamma"] + w["hnorm_beta"]
            mu, sd = out[:, :2], softplus(out[:, 2:]) + f32(1e-4)
        else:
            mu = f32(5.0) * np.tanh(out[:, :2] / f32(5.0))
            sd = softplus(out[:, 2:] + f32(RAW_INIT_STD)) + f32(1e-4)
        if self.sample:
            u = mu[None] + sd[None] * self.rng.standard_normal((100, n, 2)).astype(f32)
            a = np.tanh(u)
            # log-probability of the tanh-transformed normal, summed over the two action dimensions (constants dropped)
            logp = (-0.5 * ((u - mu[None]) / sd[None]) ** 2 - np.log(sd[None]) - np.log(np.maximum(1.0 - a * a, 1e-12))).sum(-1)
            action = a[np.argmax(logp, 0), np.arange(n)].astype(f32)
        else:
            action = np.tanh(mu).astype(f32)
        return action, dict(stoch=stoch, deter=deter, action=action)
