"""The batched chronics loader against the committed fixtures (which were cut from the same reference folders)."""
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
