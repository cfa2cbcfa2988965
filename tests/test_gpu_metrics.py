"""GPU parity of the contingency table / VI / object counts against scikit-image 0.18.3 fixtures and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vi_and_counts_match_skimage(golden):
    from iterseg_amd import metrics as gm
    d = golden("metrics_vi")
    for tag in ("a", "b", "same"):
        vi = gm.variation_of_information(d[f"{tag}.gt"], d[f"{tag}.mr"])
        assert np.allclose(vi, d[f"{tag}.vi"], rtol=1e-12, atol=1e-12), (tag, vi)
        assert list(gm.object_counts(d[f"{tag}.gt"], d[f"{tag}.mr"])[:3]) == list(d[f"{tag}.counts"])


def test_contingency_large_volume_vs_oracle():
    import torch
    from scipy import ndimage as ndi
    from oracle import metrics as om
    from iterseg_amd import metrics as gm
    rng = np.random.default_rng(8)
    f = ndi.gaussian_filter(rng.random((24, 128, 160)), (1.0, 2.0, 2.0))
    gt = ndi.label(f > np.quantile(f, 0.75))[0].astype(np.int32)
    mr = ndi.label(f > np.quantile(f, 0.72))[0].astype(np.int32)
    a, b, c = gm.contingency_table(torch.from_numpy(gt).cuda(), torch.from_numpy(mr).cuda(), cap=16)   # grows the buffer
    ra, rb, rc = om.contingency(gt, mr)
    assert np.array_equal(a, ra) and np.array_equal(b, rb) and np.array_equal(c, rc) and c.sum() == gt.size
    assert np.allclose(gm.variation_of_information(gt, mr), om.variation_of_information(gt, mr), rtol=1e-13)
    with pytest.raises(ValueError):
        gm.contingency_table(gt, mr[:-1])


def test_get_accuracy_metrics_table(golden):
    from iterseg_amd import metrics as gm
    d = golden("metrics_vi")
    gt, mr = d["a.gt"], d["a.mr"]
    full = (slice(None),) * 3
    df = gm.get_accuracy_metrics([(full, full)], gt, mr, name="m", exclude_chunks=3)
    assert len(df) == 1 and abs(df['VI: GT | Output'][0] - d["a.vi"][0]) < 1e-12
    assert int(df['Number objects (GT)'][0]) == int(d["a.counts"][0]) and df['model_name'][0] == "m"
    with pytest.raises(NotImplementedError):
        gm.get_accuracy_metrics([(full, full)], gt, mr, AP=True)
