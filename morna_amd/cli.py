"""`morna index` / `morna search` command line on the MI355X library.

Mirrors the reference's parser and dispatch (commanderson/morna
morna.py:867-1054, 1338-1484): same subcommands, flag names, defaults and output
format for the hot path, including the metadata join (-m) and the convergence
back-off loop (-c / -ch, sam input only).  Not carried over:
`junctions` and the junctions-by-sample shards (-b is accepted and ignored).

One deliberate difference in the back-off loop: when the stream ends without
convergence the reference's quiet branch prints the results of the LAST CHECKPOINT
(morna.py:1452, a NameError if no checkpoint was reached) while its verbose branch
finalises the whole query and searches again (morna.py:1421-1427); both branches
here do the latter.

    python -m morna_amd.cli index --intropolis junctions.tsv.gz -x idx -s 9662 --n-trees 10
    python -m morna_amd.cli search -x idx -q 1234 -d
    cat query.bed | python -m morna_amd.cli search -x idx -f bed --exact -d
"""
import argparse
import sys

_help_intro = """morna searches for known RNA-seq samples with exon-exon junction expression
patterns similar to those in a query sample (MI355X build of the index/search hot path).
"""


def add_search_parameters(subparser):
    subparser.add_argument('-x', '--basename', metavar='<idx>', type=str, required=True,
                           help='path to junction index basename for search')
    subparser.add_argument('-v', '--verbose', action='store_const', const=True, default=False, help='be talkative')
    subparser.add_argument('--search-k', metavar='<int>', type=int, required=False, default=100,
                           help='a larger value makes for more accurate search')
    subparser.add_argument('-f', '--format', metavar='<choice>', type=str, required=False, default='sam',
                           help='one of {sam, bed, raw}')
    subparser.add_argument('-d', '--distances', action='store_const', const=True, default=False,
                           help='include distances to nearest neighbors')
    subparser.add_argument('-m', '--metadata', action='store_const', const=True, default=False,
                           help='display results mapped to metadata')
    subparser.add_argument('-c', '--convergence-backoff', metavar='<int>', type=int, required=False, default=None,
                           help='attempt to converge on a solution with this backoff interval '
                                '(check after c, then 2c, then 4c, and stop when no change)')
    subparser.add_argument('-ch', '--checkpoint', metavar='<int>', type=int, required=False, default=0,
                           help='do not start backoff until this point (check at checkpoint, '
                                'then checkpoint + c, then continue exponential backoff)')
    subparser.add_argument('-q', '--query-id', metavar='<int>', type=int, required=False, default=None,
                           help='search for nearest neighbors to the sample already in the index with this sample id')
    subparser.add_argument('-e', '--exact', action='store_const', const=True, default=False,
                           help='exact nearest neighbor search within the morna index')
    subparser.add_argument('-r', '--results', metavar='<int>', type=int, required=False, default=20,
                           help='the number of nearest neighbor results to return')
    subparser.add_argument('-rl', '--rawlist', action='store_const', const=True, default=False,
                           help='regurgitate junction list for input sample instead of performing search')
    subparser.add_argument('--device', type=str, default='0',
                           help='HIP device ordinal; for an index built with --shards also a list, "0,1,2,3": the shards are '
                                'dealt to these devices in turn')


def build_parser():
    parser = argparse.ArgumentParser(description=_help_intro)
    subparsers = parser.add_subparsers(dest='subparser_name',
                                       help='subcommands; add "-h" or "--help" after a subcommand for its parameters')
    index_parser = subparsers.add_parser('index', help='creates a morna index')
    search_parser = subparsers.add_parser('search', help='searches a morna index')
    index_parser.add_argument('--intropolis', metavar='<file>', type=str, required=True,
                              help='path to gzipped file recording junctions across samples in intropolis format')
    index_parser.add_argument('-x', '--basename', metavar='<str>', type=str, required=False, default='morna',
                              help='basename path of junction index files to create')
    index_parser.add_argument('--features', metavar='<int>', type=int, required=False, default=3000,
                              help='dimension of feature space')
    index_parser.add_argument('--n-trees', metavar='<int>', type=int, required=False, default=200,
                              help='number of annoy trees')
    index_parser.add_argument('-s', '--sample-count', metavar='<int>', type=int, required=False, default=None,
                              help='optionally specify number of unique samples to speed indexing')
    index_parser.add_argument('-t', '--sample-threshold', metavar='<int>', type=int, required=False, default=100,
                              help='minimum number of samples in which a junction should appear')
    index_parser.add_argument('-b', '--buffer-size', metavar='<int>', type=int, required=False, default=1024,
                              help='accepted for compatibility; the per-sample junction database is out of scope')
    index_parser.add_argument('-v', '--verbose', action='store_const', const=True, default=False, help='be talkative')
    index_parser.add_argument('-m', '--metafile', metavar='<file>', type=str, required=False, default=None,
                              help='path to metadata file with sample index in first column '
                                   'and other keywords in other columns, whitespace delimited')
    index_parser.add_argument('--device', type=int, default=0, help='HIP device ordinal')
    index_parser.add_argument('--python-parse', action='store_const', const=True, default=False,
                              help='tokenise the intropolis file with the Python loop of the reference '
                                   'instead of the native pre-pass (same index either way)')
    index_parser.add_argument('--cache', metavar='<file>', type=str, required=False, default=None,
                              help='binary pre-tokenised cache of the intropolis file: reused when it matches the '
                                   'file, sample count and threshold, (re)written otherwise')
    index_parser.add_argument('--shards', metavar='<int>', type=int, required=False, default=1,
                              help='cut the samples into this many contiguous row shards, each with its own matrix + '
                                   'forest file (one per GPU); global idf and internal ids, so the shards together are '
                                   'the index.  Under torchrun with WORLD_SIZE equal to --shards every rank builds its '
                                   'own shard on its own GPU; otherwise one process builds them one after the other')
    add_search_parameters(search_parser)
    return parser


def main(argv=None, stdin=None, stdout=None):
    args = build_parser().parse_args(argv)
    stdin = stdin or sys.stdin
    stdout = stdout or sys.stdout
    if args.subparser_name == 'index':
        import os
        from .index import go_index
        rank, device = None, int(str(args.device).split(',')[0])
        if args.shards > 1 and int(os.environ.get("WORLD_SIZE", "1")) == args.shards:   # one process per GPU
            from ._lib import device_count
            rank = int(os.environ.get("RANK", "0"))
            device = int(os.environ.get("LOCAL_RANK", "0")) % device_count()   # (fewer GPUs than ranks: the ranks share them)
        go_index(args.intropolis, args.basename, args.features, args.n_trees, args.sample_count,
                 args.sample_threshold, args.buffer_size, args.verbose, args.metafile, device=device,
                 native=not args.python_parse, cache=args.cache, shards=args.shards, rank=rank)
        return 0
    if args.subparser_name != 'search':
        build_parser().print_help()
        return 2
    from .search import MornaSearch, results_output
    from .streams import junctions_from_bed_stream, junctions_from_raw_stream, junctions_from_sam_stream
    devices = [int(d) for d in str(args.device).split(',')]
    import os
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1 and os.path.exists(args.basename + ".shards.mor"):
        # torchrun with one process per shard: every rank runs this function with the same query (rank 0 reads the stream
        # and hands it over) and calls the same collectives; rank 0 prints
        import io
        import torch
        import torch.distributed as dist
        from ._lib import device_count
        n_dev = device_count()
        local = int(os.environ.get("LOCAL_RANK", "0"))
        backend = "nccl" if n_dev >= int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) else "gloo"   # RCCL wants a GPU per rank
        torch.cuda.set_device(local % n_dev)
        dist.init_process_group(backend, rank=rank, world_size=world)
        searcher = MornaSearch(basename=args.basename, device=local % n_dev, rank=rank, world=world)
        if rank != 0:
            stdout = io.StringIO()
    else:
        searcher = MornaSearch(basename=args.basename, device=devices if len(devices) > 1 else devices[0])
    try:
        return _search(args, searcher, stdin, stdout, dist, rank)
    finally:
        if dist is not None:
            searcher.annoy_index.close()
            dist.destroy_process_group()


def _search(args, searcher, stdin, stdout, dist, rank):
    from .search import results_output
    from .streams import junctions_from_bed_stream, junctions_from_raw_stream, junctions_from_sam_stream
    if args.query_id is not None:                              # morna.py:1358-1365
        if dist is not None and rank != 0:
            import contextlib
            with contextlib.redirect_stdout(stdout):           # ("querying by sample id ..." is rank 0's to print)
                results = searcher.search_member_n(args.query_id, args.results, args.search_k,
                                                   include_distances=args.distances, meta_db=args.metadata)
        else:
            results = searcher.search_member_n(args.query_id, args.results, args.search_k,
                                               include_distances=args.distances, meta_db=args.metadata)
        results_output(results, stdout)
        return 0
    if dist is not None:
        # only rank 0 has the stream: it parses it and every rank walks the same list of junctions
        box = [None]
        if rank == 0:
            gen = {"sam": junctions_from_sam_stream, "bed": junctions_from_bed_stream, "raw": junctions_from_raw_stream}[args.format]
            box[0] = list(gen(stdin))
        dist.broadcast_object_list(box, src=0)
        junction_generator = iter(box[0])
    elif args.format == "sam":
        junction_generator = junctions_from_sam_stream(stdin)
    elif args.format == "bed":
        junction_generator = junctions_from_bed_stream(stdin)
    else:
        assert args.format == "raw"
        junction_generator = junctions_from_raw_stream(stdin)
    if args.rawlist:
        for junction in junction_generator:
            stdout.write(str(junction) + "\n")
        return 0
    converge = bool(args.convergence_backoff) and args.format == 'sam'      # morna.py:1378
    backoff = args.convergence_backoff
    checkpoint = args.checkpoint
    old_results = [-1 for _ in range(args.results)]
    i = -1
    for i, junction in enumerate(junction_generator):          # morna.py:1383-1472
        string_junction = " ".join(str(_) for _ in junction[:3])
        if args.verbose and i % 1000 == 0:
            sys.stderr.write(str(i) + " junctions into query sample\r")
            sys.stderr.flush()
        if string_junction in searcher.sample_frequencies:
            searcher.update_query(junction)
        if converge and i == checkpoint:
            checkpoint += backoff
            backoff += backoff
            sys.stderr.write("\n")
            searcher.finalize_query()
            results = searcher.search_nn(args.results, args.search_k, include_distances=args.distances,
                                         meta_db=args.metadata)
            same = True
            for j, result in enumerate(results[0]):
                if not (result == old_results[j]):
                    same = False
            if same:
                if args.verbose:
                    sys.stderr.write("Converged after " + str(i) + " junctions.\n")
                results_output(results, stdout)
                return 0
            if args.verbose:
                sys.stderr.write("Not converged after " + str(i) + " junctions.\n")
                sys.stderr.write("Old results:\n" + str(old_results) + "\n")
                sys.stderr.write("New results:\n" + str(results[0]) + "\n")
            old_results = results[0]
    if converge and args.verbose:
        sys.stderr.write("No convergence after " + str(i) + " junctions, but here's results:\n")
    searcher.finalize_query()
    if args.verbose:
        sys.stderr.write("\n")
    if args.exact and not converge:
        results = searcher.exact_search_nn(args.results, include_distances=args.distances, meta_db=args.metadata)
    else:
        results = searcher.search_nn(args.results, args.search_k, include_distances=args.distances,
                                     meta_db=args.metadata)
    results_output(results, stdout)
    return 0


if __name__ == '__main__':
    sys.exit(main())
