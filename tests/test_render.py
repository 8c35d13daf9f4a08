"""`env.render(mode, agent)` of the shim as the reference's `Render` wrapper consumes it (dreamer/wrappers.py:161-195):
geometry of the two views on a known scene, then the reference's own Render class on top of the shim."""
import os

import numpy as np
import pytest

from conftest import REF
from racing_dreamer_amd.compat.racecar_gym.envs.rendering import COLORS, FRAME, render_view
from racing_dreamer_amd.track_assets import load_track, synthetic_track


def _state(*poses):
    return {aid: {"pose": np.array([x, y, 0, 0, 0, yaw])} for aid, (x, y, yaw) in zip("ABCD", poses)}


def test_birds_eye_is_north_up_and_centred_on_the_agent():
    t = load_track("columbia")
    x, y, yaw = (float(v) for v in t.centerline[40, :3])
    img = render_view(t, _state((x, y, yaw)), "A", "birds_eye")
    assert img.shape == FRAME + (3,) and img.dtype == np.uint8
    h, w = FRAME
    assert tuple(img[h // 2, w // 2]) in (tuple(COLORS[0]), tuple((COLORS[0].astype(int) * 6 // 10).astype(np.uint8)))
    # every pixel equals the track grid at the world point it stands for (16 m x 12 m window, 0.05 m per pixel)
    mpp = 12.0 / h
    for r, c in ((10, 10), (h - 5, w - 7), (60, 200), (200, 40)):
        wx, wy = x + (c + 0.5 - w / 2) * mpp, y + (h / 2 - (r + 0.5)) * mpp
        ix, iy = int(np.floor((wx - t.origin[0]) / 0.05)), int(np.floor((wy - t.origin[1]) / 0.05))
        if 0 <= ix < t.width and 0 <= iy < t.height:
            want = (40, 40, 40) if t.occ[iy, ix] else ((255, 255, 255) if t.drivable[iy, ix] else (235, 235, 235))
        else:
            want = (235, 235, 235)
        if abs(wx - x) > 0.6 or abs(wy - y) > 0.6:
            assert tuple(img[r, c]) == want, (r, c)
    # moving the agent north shifts the scene down by the same number of pixels
    img2 = render_view(t, _state((x, y + 1.0, yaw)), "A", "birds_eye")
    d = int(round(1.0 / mpp))
    a, b = img[: h - d, :5], img2[d:, :5]                        # a strip far from the car
    assert np.array_equal(a, b)


def test_follow_view_is_ego_aligned():
    t = synthetic_track(height=200, width=300, wall=8)
    x0, y0 = (float(v) for v in t.centerline[0, :2])
    h, w = FRAME
    frames = {}
    for yaw in (0.0, np.pi / 2, -2.0):
        # a second car 1.5 m straight ahead of the focus car, whatever its heading
        ahead = (x0 + 1.5 * np.cos(yaw), y0 + 1.5 * np.sin(yaw), yaw + 0.3)
        img = render_view(t, _state((x0, y0, yaw), ahead), "A", "follow")
        frames[yaw] = img
        mpp = 6.0 / h
        mine = np.all(img == COLORS[0], axis=-1) | np.all(img == (COLORS[0].astype(int) * 6 // 10).astype(np.uint8), axis=-1)
        rows, cols = np.nonzero(mine)
        # the focus car is a vertical rectangle (heading = up), 0.55 m long and 0.30 m wide, three quarters down the frame
        assert abs((rows.max() - rows.min() + 1) * mpp - 0.55) < 0.06 and abs((cols.max() - cols.min() + 1) * mpp - 0.30) < 0.06
        assert abs(cols.mean() - w / 2) < 1.5 and abs(rows.mean() - (0.75 * h - 0.175 / mpp)) < 2.0
        nose = np.all(img == (COLORS[0].astype(int) * 6 // 10).astype(np.uint8), axis=-1)
        assert np.nonzero(nose)[0].mean() < rows.mean()            # the darker front third is the upper end
        other = np.all(img == COLORS[1], axis=-1)
        orow, ocol = np.nonzero(other)
        assert abs(ocol.mean() - w / 2) < 3 and abs((rows.mean() - orow.mean()) * mpp - 1.5) < 0.1     # straight ahead = straight up
    assert not np.array_equal(frames[0.0], frames[np.pi / 2])      # the scene turned with the car
    with pytest.raises(ValueError, match="render mode"):
        render_view(t, _state((x0, y0, 0.0)), "A", "first_person")


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout")
def test_reference_render_wrapper_collects_per_agent_videos_on_the_shim(ref_wrappers):
    W = ref_wrappers
    os.chdir(os.path.join(REF, "baselines"))                      # the 4-agent scenario A..D
    videos = []
    env = W.RaceCarWrapper(W.RaceCarBaseEnv(track="columbia", task="max_progress"), agent_id="A")
    env = W.TimeLimit(W.FixedResetMode(env, mode="grid"), 40)      # (0.4 s: on columbia's 34 m wide map a bird's-eye pixel is 11 cm)
    env = W.Render(env, callbacks=[lambda v: videos.append({k: list(f) for k, f in v.items()})], follow_view=True)
    env.reset()
    done, steps = False, 0
    while not done:
        obs, rew, dones, info = env.step({a: np.array([0.6, 0.0]) for a in env.agent_ids})
        done, steps = any(dones.values()), steps + 1
    assert steps == 40 and len(videos) == 1
    v = videos[0]
    assert sorted(v) == ["birds_eye-A", "follow-A", "follow-B", "follow-C", "follow-D"]          # wrappers.py:169-173
    for k, frames in v.items():
        assert len(frames) == steps + 1                            # the reset frame + one per step (wrappers.py:178-194)
        assert all(f.shape == (240, 320, 3) and f.dtype == np.uint8 for f in frames)
    # each follow view is centred on ITS agent: the focus car (with its darker nose) sits at the same place in every
    # view, in that agent's own colour
    for k, aid in enumerate("ABCD"):
        f = v[f"follow-{aid}"][-1]
        nose = np.all(f == (COLORS[k].astype(int) * 6 // 10).astype(np.uint8), axis=-1)
        rows, cols = np.nonzero(nose)
        assert len(rows) > 20 and abs(cols.mean() - 160) < 2 and 150 < rows.mean() < 185
    # the cars drove off: the scene of the last birds-eye frame differs from the first
    assert not np.array_equal(v["birds_eye-A"][0], v["birds_eye-A"][-1])
