import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a fresh checkout has no libchebgcn.so (built artefacts are not in history): build it once, as
    # __graft_entry__.build() does, so that the ABI / host tests do not depend on the order of the driver's steps
    so = os.path.join(ROOT, 'gcn_fmri_decoding_amd', 'libchebgcn.so')
    if not os.path.exists(so):
        import shutil
        import subprocess
        if shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc'):
            r = subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'gcn_fmri_decoding_amd', 'csrc')], check=False)
            if r.returncode != 0 or not os.path.exists(so):
                # say so here: otherwise the failure only shows up later as "libchebgcn.so not found"
                pytest.exit('building libchebgcn.so failed (make exit code %d); see the compiler output above'
                            % r.returncode, returncode=3)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def csr_from(z, prefix):
    shape = tuple(int(s) for s in z[prefix + '_shape'])
    return sp.csr_matrix((z[prefix + '_data'], z[prefix + '_indices'], z[prefix + '_indptr']), shape=shape)


def assert_csr_equal(A, B, exact=True, rtol=0.0, atol=0.0):
    A, B = sp.csr_matrix(A).copy(), sp.csr_matrix(B).copy()
    A.sort_indices(); B.sort_indices()
    assert A.shape == B.shape
    assert np.array_equal(A.indptr, B.indptr)
    assert np.array_equal(A.indices, B.indices)
    if exact:
        assert np.array_equal(A.data, B.data)
    else:
        np.testing.assert_allclose(A.data, B.data, rtol=rtol, atol=atol)


@pytest.fixture(scope='session')
def golden():
    return load_golden


def assert_adam_params_close(got, ref, v_ref, step, ill, key, rel=2e-5, lr=1e-3, quantile=1.0):
    """Variables after ``step + 1`` TF-form Adam steps (lr 1e-3) against the oracle's.  The update
    lr_t * m / (sqrt(v) + eps) is ~ +-lr whenever the gradient RMS is >> eps = 1e-8 and ill-conditioned
    where it is 1e-8 .. 1e-6 (there an fp32 round-off in g changes the quotient): ``rel`` relative to
    max|ref| on the well-conditioned elements (RMS gradient > 1e-5, or exactly zero; once an element was
    ill-conditioned it stays excluded -- ``ill`` carries that mask between steps), and nothing moves by
    more than one learning rate per step anywhere."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    scale = max(np.abs(ref).max(), 1e-30)
    rms = np.sqrt(np.asarray(v_ref, np.float64) / (1 - 0.999 ** (step + 1)))
    ill[key] = ill.get(key, False) | ~((rms > 1e-5) | (rms == 0))
    well = ~ill[key]
    if well.any():
        # quantile < 1 (later steps): a ReLU flipping on one side only moves single elements by a whole update
        worst = d[well].max() if quantile >= 1.0 else np.quantile(d[well], quantile)
        assert worst <= rel * scale, 'step %d %s: rel err %.3e' % (step, key, worst / scale)
    assert d.max() <= 1.1 * lr * (step + 1), 'step %d %s: max diff %.3e' % (step, key, d.max())


def record_measured(test, **values):
    """Measured parity errors of a GPU test, appended as one JSON line to gpurun_out/parity_measured.jsonl (scratch on the
    GPU box; the round's copy is committed as profiles/rNN_parity_measured.jsonl).  A passing test prints nothing under
    ``pytest -q``: this is where the margins to the asserted bounds can be read afterwards."""
    import json
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'parity_measured.jsonl'), 'a') as f:
            f.write(json.dumps(dict(test=test, **{k: (float(v) if isinstance(v, (int, float, np.floating)) else v)
                                                  for k, v in values.items()})) + '\n')
    except OSError:
        pass
