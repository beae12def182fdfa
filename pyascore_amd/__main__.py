"""``python -m pyascore_amd [options] spec_file ident_file out_file`` -- the reference's command line
(`pyascore/__main__.py`, options of `pyascore/config.py:19-93`, same names and defaults) on the
MI355X scorer: the files are read by :mod:`pyascore_amd.ingest`, every selected PSM is scored in ONE
batched call (:func:`pyascore_amd.batch_cli.localize`) and the TSV of docs/source/cli.rst:135-180 is
written.  ``--parameter_file`` takes ``name = value`` lines ('#' starts a comment); options on the
command line override it.  ``--device`` (HIP ordinal) is the one addition."""
import argparse
import re
import sys
from datetime import datetime


def args_from_file(path):
    """``name = value`` lines -> ``["--name", "value", ...]`` (config.py:5-17)."""
    out = []
    with open(path) as f:
        for line in f:
            m = re.search(r"^(\S+)\s*=\s*(\S+)$", line.split("#")[0].strip())
            if m:
                out += ["--" + m.group(1), m.group(2)]
    return out


def build_parser():
    p = argparse.ArgumentParser(prog="pyascore_amd", description="PTM site localisation (Ascore) on an MI355X: "
                                "spectra (mzML / mzXML) + identifications (pepXML / mzIdentML / percolatorTXT / "
                                "mokapotTXT) -> Scan, LocalizedSequence, PepScore, Ascores, AltSites.")
    p.add_argument("--match_save", action="store_true",
                   help="write dump_spectra.pkl / dump_match.pkl of the last scored PSM, as the reference's loop leaves them")
    p.add_argument("--residues", type=str, default="STY", help="residues that can carry the modification")
    p.add_argument("--mod_mass", type=float, default=79.966331, help="exact mass of the modification")
    p.add_argument("--mz_error", type=float, default=0.5, help="fragment match tolerance in m/z")
    p.add_argument("--mod_correction_tol", type=float, default=1.0,
                   help="how far a reported modification mass may be from --mod_mass")
    p.add_argument("--zero_based", type=bool, default=False, help="modification positions count from 0")
    p.add_argument("--neutral_loss_groups", type=str, default="", help="comma separated residue groups (lower case: modified form)")
    p.add_argument("--neutral_loss_masses", type=str, default="", help="one loss mass per group")
    p.add_argument("--static_mod_groups", type=str, default="C", help="comma separated residue groups with a constant modification")
    p.add_argument("--static_mod_masses", type=str, default="57.021464", help="one mass per static group")
    p.add_argument("--fragment_types", type=str, default="by", help="ion types to score, of bcyzZ")
    p.add_argument("--max_fragment_charge", type=int, default=5, help="upper limit of the fragment charge (also PSM charge - 1)")
    p.add_argument("--hit_depth", type=int, default=1, help="PSMs taken per scan; negative = all")
    p.add_argument("--parameter_file", type=str, default="", help="file of 'name = value' lines")
    p.add_argument("--spec_file_type", type=str, default="mzML", help="mzML or mzXML")
    p.add_argument("--ident_file_type", type=str, default="pepXML", help="pepXML, mzIdentML, percolatorTXT or mokapotTXT")
    p.add_argument("--device", type=int, default=None, help="HIP device ordinal (default: LOCAL_RANK or 0)")
    p.add_argument("spec_file", type=str)
    p.add_argument("ident_file", type=str)
    p.add_argument("out_file", type=str)
    return p


def validate_args(args):
    """`__main__.py:48-65`."""
    for aa in args.residues:
        if aa not in "ncACDEFGHIKLMNOPQRSTUVWY":
            raise ValueError("The residue inputed, {}, is not allowed.".format(aa))
    for frag in args.fragment_types:
        if frag not in "cbyzZ":
            raise ValueError("The fragment type inputed, {}, is not allowed.".format(frag))
    if args.max_fragment_charge < 1:
        raise ValueError("The max fragment charge must be greater than or equal to 1")


def parse_args(argv):
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.parameter_file:
        args = parser.parse_args(args_from_file(args.parameter_file) + list(argv))
    validate_args(args)
    return args


def static_mods_of(args):
    mods = {}
    for group, mass in zip(args.static_mod_groups.split(","), args.static_mod_masses.split(",")):
        mods.update({aa: float(mass) for aa in group})
    return mods


def run(args, log=print):
    from . import batch_cli, ingest
    from .ascore import PyAscore
    stamp = lambda: datetime.now().strftime("%m/%d/%y %H:%M:%S")
    log("{} -- Ascore Started".format(stamp()))
    log("{} -- Reading spectra from: {}".format(stamp(), args.spec_file))
    spectra = ingest.SpectraParser(args.spec_file, args.spec_file_type).to_dict()
    log("{} -- Reading identifications from: {}".format(stamp(), args.ident_file))
    static = static_mods_of(args)
    known = dict(ingest.COMMON_MODS)                       # `__main__.py:30-35`
    known.update({aa: args.mod_mass for aa in args.residues})
    known.update(static)
    psms = sorted(ingest.IdentificationParser(args.ident_file, args.ident_file_type,
                                              ingest.MassCorrector(mod_mass_dict=known), static_mods=static).to_list(),
                  key=lambda p: p["scan"])
    log("{} -- Anlyzing PSMs".format(stamp()))
    kw = {} if args.device is None else {"device": args.device}
    ascore = PyAscore(bin_size=100.0, n_top=10, mod_group=args.residues, mod_mass=args.mod_mass,
                      mz_error=args.mz_error, fragment_types=args.fragment_types, **kw)
    if args.neutral_loss_groups and args.neutral_loss_masses:
        for group, mass in zip(args.neutral_loss_groups.split(","), args.neutral_loss_masses.split(",")):
            ascore.add_neutral_loss(group, float(mass))
    rows = batch_cli.localize(ascore, psms, spectra, args.residues, args.mod_mass, args.hit_depth,
                              args.max_fragment_charge, args.mod_correction_tol, args.zero_based,
                              match_save=args.match_save, log=lambda m: log("{} -- {}".format(stamp(), m)))
    batch_cli.write_tsv(rows, args.out_file)
    log("{} -- Ascore Completed".format(stamp()))
    return rows


def main(argv=None):
    run(parse_args(sys.argv[1:] if argv is None else argv))


if __name__ == "__main__":
    main()
