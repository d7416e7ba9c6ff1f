"""`render()`'s return value (gaussian_renderer.RenderPackage): a dict whose "visibility_filter" (= radii > 0, reference
gaussian_renderer/__init__.py:118-121) is computed on first use.  Whatever way a caller looks at the dict, the key is there."""
import torch

from gaussian_renderer import RenderPackage


def _pkg():
    return RenderPackage({"render": 1, "viewspace_points": 2, "radii": torch.tensor([0, 3, 0, 1]), "depth": 4})


def test_key_access_and_laziness():
    p = _pkg()
    assert not dict.__contains__(p, "visibility_filter")          # nothing computed yet
    assert p["render"] == 1 and not dict.__contains__(p, "visibility_filter")
    assert p["visibility_filter"].tolist() == [False, True, False, True]
    assert p["visibility_filter"] is p["visibility_filter"]       # computed once


def test_every_enumeration_sees_the_key():
    keys = {"render", "viewspace_points", "visibility_filter", "radii", "depth"}
    assert set(_pkg().keys()) == keys and set(dict(_pkg())) == keys and set({**_pkg()}) == keys
    assert set(k for k in _pkg()) == keys and len(_pkg()) == 5 and "visibility_filter" in _pkg()
    assert set(k for k, _ in _pkg().items()) == keys and _pkg().get("visibility_filter") is not None
    assert _pkg().copy()["visibility_filter"].dtype == torch.bool
