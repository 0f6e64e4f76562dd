"""CPU: the conservative cull rules -- the alpha >= 1/255 box that bounds the tile rectangle of GSR_FLAG_TILE_BOUNDS_ALPHA (K1)
and the blend kernels' ellipse-against-rectangle test (gsr_blend.hip: can_touch_quad) -- replayed in binary32 against the
reference's per-pixel evaluation (forward.cu:335-344) on random Gaussians that include needles -- tools/cull_replay.py.  ADVICE r01: the round-1 rule dropped pixels the
reference blends when det(conic) cancels; the current rule must not, and must still cull the well-conditioned ones."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_cull_box_is_conservative_for_needles():
    import cull_replay

    refused, fails, worst = cull_replay.replay(4000, old=False, seed=1)
    assert fails == 0 and worst == 0.0
    assert 0 < refused < 2000  # ill-conditioned conics are kept, the rest is still culled by the box
    # the replay has teeth: it finds the failures of the round-1 rule
    _, old_fails, old_worst = cull_replay.replay(4000, old=True, seed=1)
    assert old_fails > 20 and old_worst > 1.0 / 255.0


def test_kernel_source_uses_the_replayed_rule():
    """The constants replayed above are the ones in the kernels (both sites)."""
    for f in ("gsr_blend.hip", "gsr_preprocess.hip"):
        src = open(os.path.join(ROOT, "gaussianeditor_amd", "csrc", f)).read()
        assert "1e-3f * xz" in src and "* 1.001f" in src, f


def test_ellipse_rectangle_cull_is_conservative():
    """can_touch_quad's exact ellipse-against-rectangle test (round 2), replayed in binary32: no (Gaussian, rectangle) pair
    it culls contains a pixel the reference would blend."""
    import cull_replay

    culled, fails, worst, kept = cull_replay.replay_rect(600, seed=2, rects_per=30)
    assert culled > 2000 and kept > 2000
    assert fails == 0, (fails, worst)
