"""Option sets of the MFCC front-end fixtures (tests/golden/fe.npz), shared by make_golden.py (which runs the
unmodified reference with them) and the tests (which map them onto s3a_fe_params_t / s3o_fe_params_t)."""

FE_CHAN3 = ["-samprate", "11025", "-frate", "105", "-wlen", "0.024", "-alpha", "0.97", "-ncep", "13", "-nfft", "512",
            "-nfilt", "36", "-upperf", "5400", "-lowerf", "130", "-input_endian", "little"]
# name -> (input, options): the option sets of the reference's own sphinx_fe regression tests
# (sphinxbase/test/regression/test-sphinx_fe*.sh) on its chan3.raw, and others on pocketsphinx's goforward.raw
FE_CASES = {
    "chan3": ("chan3", FE_CHAN3),
    "chan3_logspec": ("chan3", FE_CHAN3 + ["-logspec", "1"]),
    "chan3_smoothspec": ("chan3", FE_CHAN3 + ["-smoothspec", "1"]),
    "goforward": ("goforward", []),
    "goforward_dct_lifter_dc": ("goforward", ["-transform", "dct", "-lifter", "22", "-remove_dc", "yes"]),
    "goforward_htk_plain": ("goforward", ["-transform", "htk", "-round_filters", "no", "-unit_area", "no", "-alpha", "0"]),
    "goforward_8k_doublebw": ("goforward", ["-samprate", "8000", "-nfft", "256", "-nfilt", "31", "-lowerf", "200",
                                            "-upperf", "3500", "-doublebw", "yes"]),
    "goforward_logspec_lifter": ("goforward", ["-logspec", "1", "-lifter", "10"]),
}
FE_SHORT = (0, 1, 100, 409, 410, 411, 569, 570, 571, 730, 1000)     # sample counts around the framing edges


def fe_params(opts):
    """reference command-line options -> the fields of the params structs that differ from the defaults"""
    out = {}
    it = iter(opts)
    for k in it:
        v = next(it)
        k = k.lstrip("-")
        if k == "input_endian":
            continue
        if k == "transform":
            out[k] = {"legacy": 0, "dct": 1, "htk": 2}[v]
        elif k == "logspec":
            out[k] = 1 if v in ("1", "yes") else 0
        elif k == "smoothspec":
            if v in ("1", "yes"):
                out["logspec"] = 2
        elif k in ("samprate", "wlen", "alpha", "lowerf", "upperf"):
            out[k] = float(v)
        elif k in ("remove_dc", "round_filters", "unit_area", "doublebw"):
            out[k] = 1 if v in ("1", "yes") else 0
        else:
            out[k] = int(v)
    return out
