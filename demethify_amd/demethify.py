"""``demethify`` command line: same flags, defaults, input formats and output files as the
reference's demethify/demethify.py, with the solver running on the GPU.

Multi-GPU: launch one process per GPU, e.g.
    python -m torch.distributed.run --nproc-per-node 8 -m demethify_amd --methfreq ... --restart 64
Restarts, bootstrap replicates and model-selection candidates are then sharded over the ranks
(demethify_amd/shard.py); rank 0 writes the output files.

Deliberate differences from upstream, all flagged at run time:
  * ``--restart r``: upstream re-runs the SAME seed r times; here restart k > 0 uses seed + k
    (restart 0 is bit-compatible), SURVEY.md section 8b.
  * ``--plot`` is ignored with a note and ``--init SVD|ICA`` exits with a message (outside this build's scope).
  * ``--ic NAME [n [lo hi]]``: upstream sweeps the hard-coded range 1..25 (ic.py:171); two more values after the
    restart count restrict the sweep to lo..hi unknown types (``--ic BIC 5 2 12``).  Without them: 1..25, as upstream.
"""
from __future__ import annotations

import argparse
import os
import sys
import warnings
from time import time

import numpy as np
import pandas as pd

from . import tables

logo = r"""
    ____                      __  __    _ ____
   / __ \___  ____ ___  ___  / /_/ /_  (_) __/_  __
  / / / / _ \/ __ `__ \/ _ \/ __/ __ \/ / /_/ / / /
 / /_/ /  __/ / / / / /  __/ /_/ / / / / __/ /_/ /
/_____/\___/_/ /_/ /_/\___/\__/_/ /_/_/_/  \__, /
                                          /____/    (MI355X build)
"""


def build_parser():
    """Flag surface of demethify/demethify.py:27-45 (same names, nargs, types and defaults)."""
    parser = argparse.ArgumentParser(description="DeMethify - Partial reference-based Methylation Deconvolution")
    parser.add_argument('--methfreq', nargs='+', type=str, required=True, help='Methylation frequency file path (values between 0 and 1)')
    parser.add_argument('--ref', nargs='?', type=str, help='Methylation reference matrix file path')
    parser.add_argument('--iterations', nargs=2, type=int, help='Numbers of iterations for outer and inner loops (default without purity = 10000, 20, with purity= 100, 500)')
    parser.add_argument('--nbunknown', nargs=1, type=int, help="Number of unknown cell types to estimate ")
    parser.add_argument('--purity', nargs='+', type=float, help="The purities of the samples in percent [0,100], if known")
    parser.add_argument('--termination', nargs=1, type=float, default=1e-2, help='Termination condition for cost function (default = 1e-2)')
    parser.add_argument('--init', nargs="?", default='uniform_', help='Initialisation option, the default is uniform_, and the options are: uniform, uniform_, beta, SVD, ICA. ')
    parser.add_argument('--outdir', nargs='?', required=True, help='Output directory')
    parser.add_argument('--fillna', action="store_true", help='Replace every NA by 0 in the given data')
    parser.add_argument('--ic', nargs="+", help='Select number of unknown cell types by minimising a criterion (AIC, BIC, CCC, BCV, minka)')
    parser.add_argument('--confidence', nargs=2, type=int, help='Outputs bootstrap confidence intervals, takes confidence level and boostrap iteration numbers as input.')
    parser.add_argument('--plot', action="store_true", help='Plot cell type proportions estimates for each sample, eventually with confidence intervals. ')
    parser.add_argument('--restart', nargs=1, type=int, help='Number of random restarts among which to select the one with the lowest cost/highest loglikelihood')
    parser.add_argument('--seed', nargs=1, type=int, default=1, help='Set a seed integer number for random number generation for reproducibility. ')
    parser.add_argument('--noprint', action="store_true", help='Doesnt show the logo.')
    parser.add_argument('--bedmethyl', action='store_true', help="Flag to indicate that the input will be bedmethyl files, modkit style")
    return parser


def read_inputs(args, parsed=None):
    """demethify.py:102-143: bedmethyl (tab separated, percent) or csv (fractions) input.  The per-sample files
    are parsed in parallel (and 1 / world of them per rank when distributed): demethify_amd/tables.py."""
    ref, header = None, []
    sep = '\t' if args.bedmethyl else ','
    if args.ref:
        table = pd.read_csv(args.ref, sep=sep)
        if args.bedmethyl:
            table = table.iloc[:, 3:]
        if args.fillna:
            table = table.fillna(0)
        header = list(table.columns)
        ref = table.values
    meth_f, counts = tables.read_samples(args.methfreq, bool(args.bedmethyl), bool(args.fillna), parsed=parsed)
    return ref, header, meth_f, counts


def _init_distributed():
    """One process per GPU when launched through torch.distributed.run; no-op otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0
    import torch
    import torch.distributed as dist

    device = int(os.environ.get("DEMETHIFY_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("DEMETHIFY_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(device)
    if backend == "nccl":  # RCCL over xGMI: one rank per GPU
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
    else:  # gloo: CPU collectives (tests; several ranks may then share one GPU via DEMETHIFY_DEVICE)
        dist.init_process_group(backend="gloo")
    return dist.get_rank()


def main(argv=None):
    warnings.filterwarnings("ignore")
    args = build_parser().parse_args(argv)

    # ---- post-parse normalisation, demethify.py:51-100
    if args.restart is None:
        args.restart = 1
    else:
        args.restart = args.restart[0]
    if not args.iterations:
        args.iterations = [100, 500] if args.purity else [10000, 20]
    if isinstance(args.termination, list):
        args.termination = args.termination[0]
    purity = None
    if args.purity:  # demethify.py:69-77
        purity = np.array(args.purity)
        if np.any((purity >= 0) & (purity <= 1)):
            print("Purity is between 0 and 1, are you sure that it's a percentage?")
        elif np.any((purity < 0) & (purity > 100)):
            sys.stderr.write("Error: Invalid value for purity, not within [0,100] bounds.")
            sys.exit(1)
        purity = 1 - (purity / 100.0)
    nb_r = 5
    ic_range = None
    if args.ic:
        if args.nbunknown:
            sys.stderr.write("Error: --ic cannot be used with --nbunknown.\n")
            sys.exit(1)
        if len(args.ic) > 1:
            nb_r = int(args.ic[1])
        if len(args.ic) == 4:  # extension: candidate range lo..hi (upstream ignores anything after the count)
            lo, hi = int(args.ic[2]), int(args.ic[3])
            if lo < 1 or hi < lo:
                sys.stderr.write("Error: --ic NAME n lo hi needs 1 <= lo <= hi.\n")
                sys.exit(1)
            ic_range = range(lo, hi + 1)
        elif len(args.ic) > 2:
            sys.stderr.write("Error: --ic takes NAME [n_restarts [lo hi]].\n")
            sys.exit(1)
        args.ic = args.ic[0]
    if args.init in ("SVD", "ICA"):
        sys.stderr.write(f"Error: --init {args.init} (one-shot LAPACK initialiser, demethify/init_func.py) is not part "
                         "of this build; use uniform_, uniform or beta.\n")
        sys.exit(1)
    if args.plot:
        sys.stderr.write("Note: --plot is ignored, plotting (seaborn/colorcet) is outside this build's scope.\n")

    # this rank's share of the sample files is parsed BEFORE the process group / the GPU come up: the parser's
    # worker pool is forked, and a fork must not start from a process that holds a HIP runtime and RCCL threads
    parsed = tables.parse_share(args.methfreq, bool(args.bedmethyl), bool(args.fillna))
    rank = _init_distributed()
    if not args.noprint and rank == 0:
        print(logo)
    outdir = os.path.join(os.getcwd(), args.outdir)
    if rank == 0 and not os.path.exists(outdir):
        print(f'Creating directory {outdir} to store results')
        os.mkdir(outdir)
    if args.nbunknown is None:
        args.nbunknown = [0]

    ref, header, meth_f, counts = read_inputs(args, parsed)
    del parsed
    args.methfreq = [name.split("/")[-1] for name in args.methfreq]

    # imported late so that ``--help`` and argument errors work on a box without the GPU library
    from . import _lib as L
    from . import shard, staging
    from .bootstrap import bt_ci
    from .deconvolution import _init_unsupervised, init_BSSMF_md, init_BSSMF_md_p
    from .device import Problem, Solver, get_context
    from .ic import evaluate_best_ic
    from .init_func import wls_intercept

    time_start = time()
    n_u = args.nbunknown[0]
    ic_n_u = None

    if args.confidence:
        bt_ci(args.confidence[0], args.confidence[1], n_u, meth_f, counts, ref, args.init, args.iterations[0],
              args.iterations[1], args.termination, header, outdir, args.methfreq, args.purity, args.seed,
              materialize=False)

    if args.ic:
        ref_estimate, proportions, ic_n_u, _scores = evaluate_best_ic(
            meth_f, ref, counts, args.init, args.ic, args.seed, iter1=args.iterations[0], iter2=args.iterations[1],
            tol=args.termination, n_restarts=nb_r, n_u_values=ic_range)
        unknown_header = ["unknown_cell_" + str(i + 1) for i in range(ic_n_u)]
        header = header + unknown_header
    elif (not args.ref) or (n_u > 0 and meth_f.shape[1] >= 1):
        if not args.ref and n_u < 1:
            sys.exit(f'Invalid number of unknown value! : "{args.nbunknown}" ')
        unsupervised = not args.ref
        K = (0 if unsupervised else ref.shape[1]) + n_u
        if args.restart > 1 and rank == 0:
            print(f"restart k > 0 uses seed + k (upstream repeats the same seed); {args.restart} restarts")
        with Problem(get_context(), meth_f, counts, None if unsupervised else ref) as problem:
            if args.restart > 1:
                staging.reserve(((meth_f.shape[0], n_u), (K, meth_f.shape[1])), count=2)

            def prepare(k):
                # the restart's initialisation (legacy-numpy draws, NNLS), drawn and uploaded one restart ahead of the
                # GPU by a worker thread (staging.Prefetcher)
                seed_k = shard.restart_seed(args.seed, k)
                if unsupervised:
                    u0, a0 = _init_unsupervised(args.init, meth_f, n_u, seed_k)
                elif purity is not None:
                    u0, _, a0 = init_BSSMF_md_p(args.init, meth_f, counts, ref, n_u, purity, rb_alg=wls_intercept,
                                                seed=seed_k, _stack=False)
                else:
                    u0, _, a0 = init_BSSMF_md(args.init, meth_f, counts, ref, n_u, rb_alg=wls_intercept, seed=seed_k, _stack=False)
                if args.restart > 1:
                    return staging.to_device((u0, a0), problem.ctx)
                return u0, a0

            def solve_one(k, best_cost, prepared):
                u0, a0 = prepared
                mode = L.DMF_MODE_UNSUPERVISED if unsupervised else L.DMF_MODE_PARTIAL
                s = Solver(problem, u0, a0, mode)
                try:
                    if purity is not None and not unsupervised:
                        s.set_purity(purity)
                    s.step(args.iterations[0], args.iterations[1], args.termination)
                    s.cost_begin()  # cost_f_w recomputed per restart (demethify.py:169,199): taken while the next one is set up
                except BaseException:
                    s.close()
                    raise
                return s

            def solve_end(s, best_cost):
                with s:
                    cost = s.cost_end()
                    if not cost < best_cost and args.restart > 1:
                        return None, None, cost  # cannot win (strict '<', demethify.py:170,200): stays on the device
                    u, alpha, _, _ = s.get()
                return u, alpha, cost

            ref_estimate, proportions, _best, _costs = shard.sharded_restarts(
                args.restart, solve_one, ((meth_f.shape[0], n_u), (K, meth_f.shape[1])), prepare=prepare, solve_end=solve_end)
        unknown_header = ["unknown_cell_" + str(i + 1) for i in range(n_u)]
        header = unknown_header if unsupervised else header + unknown_header
    elif n_u == 0 and meth_f.shape[1] >= 1:
        ref_estimate = None
        proportions = np.concatenate(
            [wls_intercept(counts[:, k:k + 1] * meth_f[:, k:k + 1], counts[:, k:k + 1], ref)
             for k in range(meth_f.shape[1])], axis=1)
    else:
        sys.exit(f'Invalid number of unknown value! : "{args.nbunknown}" ')

    time_tot = time() - time_start

    if rank == 0:
        if ref_estimate is not None:
            pd.DataFrame(ref_estimate).to_csv(outdir + '/methylation_profile_estimate.csv', index=False,
                                              header=unknown_header)
        table = pd.DataFrame(proportions)
        table.index = header
        table.columns = args.methfreq
        table.index.name = "Cell types"
        table.to_csv(outdir + '/celltypes_proportions.csv', index=True)
        print("All demethified! Results in " + outdir)
        with open(os.path.join(outdir, 'log.log'), "w+") as f:
            f.write("Total execution time = " + str(time_tot) + " s" + '\n')
            if args.ic:
                f.write("Number of unknowns that minimises " + args.ic + " : " + str(ic_n_u))


if __name__ == "__main__":
    main()
