"""Host-side restatements of device logic that can be checked without a GPU."""


def test_quick_inclusion_radii_never_contradict_the_full_test():
    """lane_quick's certain-accept / certain-reject radii (bwgr_amd/csrc/sweep.hip.h) against lane_accept's float arithmetic, both
    restated in numpy (tools/quick_accept_check.py): residual dots straddling the radii at relative distances 1e-15 .. 1e-6 never get
    a decision the full test does not give, and random residual dots are almost never left to the full test."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("quick_accept_check", os.path.join(os.path.dirname(__file__), "..", "tools", "quick_accept_check.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    for alt in (False, True):
        bad, und, tot = mod.run(3000, 3, alt)
        assert bad == 0 and und <= 2 and tot > 80000
