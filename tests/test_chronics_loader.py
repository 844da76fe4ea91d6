"""The batched chronics loader against the REFERENCE's own readers (grid2op.Chronics.GridStateFromFile / Multifolder, imported
from the read-only checkout) and against the committed fixtures."""
import os

import numpy as np
import pytest

from conftest import REFERENCE, have_reference

pytestmark = pytest.mark.skipif(not have_reference(), reason="reference data folders not available (GPU box)")


@pytest.mark.parametrize("env,scen", [("l2rpn_case14_sandbox", "0000"), ("rte_case5_example", "00"),
                                       ("l2rpn_neurips_2020_track1", "Scenario_august_dummy"),
                                       ("l2rpn_wcci_2022_dev", "2050-02-14_0")])
def test_loader_matches_fixture(env, scen, load_model, load_npz):
    from grid2op_amd.chronics import chronics_table, load_chronics_folder
    m = load_model(env)
    base = os.path.join(REFERENCE, "grid2op", "data", env)
    ch = load_chronics_folder(os.path.join(base, "chronics", scen), m, prods_charac=os.path.join(base, "prods_charac.csv"),
                              max_rows=600)
    fx = load_npz(f"{env}.chronics.npz")
    for k in ("load_p", "load_q", "prod_p"):
        assert np.array_equal(ch[k], fx[k]), k
    if "prod_v" in fx:
        assert np.array_equal(ch["prod_v"], fx["prod_v"])
    else:
        assert ch["prod_v"].shape == ch["prod_p"].shape and (ch["prod_v"] > 0).all()
    tab = chronics_table(ch)
    assert tab.dtype == np.float32 and tab.shape[1] == 2 * m.n_load + 2 * m.n_gen


def test_name_mapping(load_model):
    """names_chronics_to_backend (environment.py:431-437): file columns renamed before matching."""
    from grid2op_amd.chronics import load_chronics_folder
    m = load_model("l2rpn_case14_sandbox")
    folder = os.path.join(REFERENCE, "grid2op", "data", "l2rpn_case14_sandbox", "chronics", "0000")
    ident = {"loads": {str(n): str(n) for n in m.name_load}, "prods": {str(n): str(n) for n in m.name_gen}}
    a = load_chronics_folder(folder, m, max_rows=10)
    b = load_chronics_folder(folder, m, names_chronics_to_backend=ident, max_rows=10)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    with pytest.raises(KeyError):
        load_chronics_folder(folder, m, names_chronics_to_backend={"loads": {str(m.name_load[0]): "nope"}}, max_rows=3)


def _reference_multifolder(env, m):
    import test_backend_conformance  # noqa: F401  (puts the reference + the pandapower stand-in on sys.path)
    from grid2op.Chronics import GridStateFromFile, Multifolder
    mf = Multifolder(path=os.path.join(REFERENCE, "grid2op", "data", env, "chronics"), gridvalueClass=GridStateFromFile, max_iter=-1,
                     chunk_size=None)
    mf.initialize([str(x) for x in m.name_load], [str(x) for x in m.name_gen], [str(x) for x in m.name_line],
                  [str(x) for x in m.name_sub], names_chronics_to_backend=None)
    return mf


@pytest.mark.parametrize("env", ["l2rpn_case14_sandbox", "l2rpn_neurips_2020_track1", "l2rpn_wcci_2022_dev", "rte_case5_example",
                                 "educ_case14_storage"])
def test_multifolder_equals_the_reference_readers(env, load_model):
    """Scenario order, every row of load_p / load_q / prod_p / prod_v and the maintenance / hazards tables, as
    ``Multifolder`` + ``GridStateFromFile`` (grid2op/Chronics/multiFolder.py, gridStateFromFile.py) decode them."""
    from grid2op_amd.chronics import chronics_table, load_chronics_multifolder
    m = load_model(env)
    base = os.path.join(REFERENCE, "grid2op", "data", env)
    names, ch = load_chronics_multifolder(os.path.join(base, "chronics"), m, prods_charac=os.path.join(base, "prods_charac.csv"),
                                          truncate=True)
    mf = _reference_multifolder(env, m)
    assert names == [os.path.basename(p) for p in mf.subpaths]                 # sorted scenario order
    T = ch["load_p"].shape[1]
    for k, name in enumerate(names):
        mf.tell_id(k - 1 if k > 0 else len(names) - 1)                          # Multifolder.next_chronics advances to id + 1
        mf.next_chronics()
        mf.initialize([str(x) for x in m.name_load], [str(x) for x in m.name_gen], [str(x) for x in m.name_line],
                      [str(x) for x in m.name_sub], names_chronics_to_backend=None)
        d = mf.data
        assert os.path.basename(d.path) == name
        for key in ("load_p", "load_q", "prod_p"):
            assert np.array_equal(ch[key][k], getattr(d, key)[:T].astype(np.float32)), (name, key)
        if d.prod_v is not None:
            assert np.array_equal(ch["prod_v"][k], d.prod_v[:T].astype(np.float32)), name
        for key in ("maintenance", "hazards"):
            ref = getattr(d, key)
            if ref is not None and key in ch:
                assert np.array_equal(ch[key][k].astype(bool), np.asarray(ref)[:T].astype(bool)), (name, key)
            elif ref is not None:
                assert not np.asarray(ref).any()
    tab = chronics_table(ch)
    assert tab.shape == (len(names), T, 2 * m.n_load + 2 * m.n_gen) and tab.dtype == np.float32
    if env == "l2rpn_neurips_2020_track1":
        assert "maintenance" in ch and ch["maintenance"].sum() > 0


def test_forecast_tables_equal_the_reference_reader(load_model):
    """``GridStateFromFileWithForecasts`` (grid2op/Chronics/gridStateFromFileWithForecasts.py): the *_forecasted tables that
    ``obs.simulate`` injects."""
    import test_backend_conformance  # noqa: F401
    from grid2op.Chronics import GridStateFromFileWithForecasts
    from grid2op_amd.chronics import load_chronics_folder
    env, scen = "l2rpn_case14_sandbox", "0000"
    m = load_model(env)
    folder = os.path.join(REFERENCE, "grid2op", "data", env, "chronics", scen)
    ch = load_chronics_folder(folder, m, forecasts=True)
    ref = GridStateFromFileWithForecasts(path=folder, max_iter=-1, chunk_size=None)
    ref.initialize([str(x) for x in m.name_load], [str(x) for x in m.name_gen], [str(x) for x in m.name_line],
                   [str(x) for x in m.name_sub], names_chronics_to_backend=None)
    for key, attr in (("load_p_forecasted", "load_p_forecast"), ("load_q_forecasted", "load_q_forecast"),
                      ("prod_p_forecasted", "prod_p_forecast"), ("prod_v_forecasted", "prod_v_forecast")):
        got, want = ch[key], getattr(ref, attr)
        assert got.shape == want.shape and np.array_equal(got, want.astype(np.float32)), key
