"""Environment loader for the Go path: flags -> GameOptions / ContextOptions -> GCWrapper.

Counterpart of the reference's ``go/game.py`` ``Loader`` (go/game.py:17-102) without the
rlpytorch ``ArgsProvider`` machinery (plain ``argparse``): same flag names and defaults, same
batch descriptors per mode (go/game.py:64-93), `--use_mcts` and the `--mcts_*` flags of
elf/context_utils.py:16-69, same ``params`` (go/game_context.h:56-65).
Extra labels (``--additional_labels id,seq,...``, elf/more_labels.py) are appended to every
descriptor's input keys.
"""
import argparse

from . import go_game as go
from .elf_wrapper import GCWrapper


def add_arguments(parser=None):
    p = parser or argparse.ArgumentParser()
    # elf/context_utils.py:16-44
    p.add_argument("--num_games", type=int, default=1024)
    p.add_argument("--batchsize", type=int, default=128)
    p.add_argument("--T", type=int, default=1)
    p.add_argument("--num_collectors", type=int, default=1)
    p.add_argument("--verbose_comm", action="store_true")
    p.add_argument("--verbose_collector", action="store_true")
    p.add_argument("--wait_per_group", action="store_true")
    p.add_argument("--mcts_threads", type=int, default=0)
    p.add_argument("--mcts_rollout_per_thread", type=int, default=1)
    p.add_argument("--mcts_verbose", action="store_true")
    p.add_argument("--mcts_save_tree_filename", default="")
    p.add_argument("--mcts_verbose_time", action="store_true")
    p.add_argument("--mcts_use_prior", action="store_true")
    p.add_argument("--mcts_pseudo_games", type=int, default=0)
    p.add_argument("--mcts_pick_method", default="most_visited")
    # go/game.py:24-36
    p.add_argument("--actor_only", action="store_true")
    p.add_argument("--list_file", default="./train.lst")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--data_aug", type=int, default=-1, help="0-7, -1 = random")
    p.add_argument("--ratio_pre_moves", type=float, default=0.0)
    p.add_argument("--start_ratio_pre_moves", type=float, default=0.5)
    p.add_argument("--num_games_per_thread", type=int, default=5)
    p.add_argument("--move_cutoff", type=int, default=-1)
    p.add_argument("--mode", default="online", help="online | selfplay | offline")
    p.add_argument("--use_mcts", action="store_true")
    p.add_argument("--gpu", type=int, default=None)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--num_future_actions", type=int, default=3)
    p.add_argument("--additional_labels", default=None, help="comma-separated extra input keys")
    # extensions of this build
    p.add_argument("--ply_cap", type=int, default=0)
    p.add_argument("--end_on_two_passes", action="store_true")
    p.add_argument("--signal_restart", action="store_true")
    p.add_argument("--preset_zero_planes", action="store_true",
                   help="the batch tensors are zero-filled once (they are, by GCWrapper): never write the 11 constant planes")
    p.add_argument("--legal_mask", action="store_true", help="add the `legal` key to the actor inputs")
    p.add_argument("--board_id_base", type=int, default=0,
                   help="global id of game 0 (rank * num_games in a sharded job): keys the on-device RNG")
    return p


def descriptors(args):
    """go/game.py:64-93."""
    desc = {}
    actor_in = ["s"] + (["legal"] if getattr(args, "legal_mask", False) else [])
    if args.mode == "online":
        desc["human_actor"] = dict(batchsize=args.batchsize, input=dict(T=1, keys=set(["s"])),
                                   reply=dict(T=1, keys=set(["pi", "a"])), name="human_actor")
        desc["actor"] = dict(batchsize=args.batchsize, input=dict(T=1, keys=set(actor_in)),
                             reply=dict(T=1, keys=set(["pi", "V", "a"])), name="actor")
    elif args.mode == "selfplay":
        desc["actor"] = dict(batchsize=args.batchsize, input=dict(T=1, keys=set(actor_in)),
                             reply=dict(T=1, keys=set(["pi", "V"])), name="actor", timeout_usec=10)
    else:
        desc["train"] = dict(batchsize=args.batchsize, input=dict(T=args.T, keys=set(["s", "offline_a"])), reply=None)
    if args.additional_labels:
        extra = [k for k in args.additional_labels.split(",") if k]
        for v in desc.values():
            v["input"]["keys"].update(extra)
    return desc


def initialize(args):
    co = go.ContextOptions()
    co.num_games = args.num_games
    co.T = args.T
    co.num_collectors = args.num_collectors
    co.verbose_comm = args.verbose_comm
    co.verbose_collector = args.verbose_collector
    co.wait_per_group = args.wait_per_group
    co.max_num_threads = args.mcts_threads            # elf/context_utils.py:46-69
    mcts = co.mcts_options
    mcts.num_threads = args.mcts_threads
    mcts.num_rollout_per_thread = args.mcts_rollout_per_thread
    mcts.verbose = args.mcts_verbose
    mcts.verbose_time = args.mcts_verbose_time
    mcts.save_tree_filename = args.mcts_save_tree_filename
    mcts.use_prior = args.mcts_use_prior
    mcts.pseudo_games = args.mcts_pseudo_games
    mcts.pick_method = args.mcts_pick_method
    opt = go.GameOptions()
    opt.seed = args.seed
    opt.list_filename = args.list_file
    opt.mode = args.mode
    opt.use_mcts = args.use_mcts                      # go/game.py:34,51
    opt.verbose = args.verbose
    opt.data_aug = args.data_aug
    opt.ratio_pre_moves = args.ratio_pre_moves
    opt.start_ratio_pre_moves = args.start_ratio_pre_moves
    opt.move_cutoff = args.move_cutoff
    opt.num_games_per_thread = args.num_games_per_thread
    opt.num_future_actions = args.num_future_actions
    opt.gpu = args.gpu if args.gpu is not None else 0
    opt.ply_cap = args.ply_cap
    opt.end_on_two_passes = args.end_on_two_passes
    opt.signal_restart = args.signal_restart
    opt.preset_zero_planes = getattr(args, "preset_zero_planes", False)
    opt.board_id_base = args.board_id_base
    if getattr(args, "verbose", False):               # the reference prints these unconditionally (go/game.py:46,59-62)
        co.print()
    GC = go.GameContext(co, opt)
    params = GC.GetParams()
    if getattr(args, "verbose", False):
        print("Version: ", GC.Version())
        print("Num Actions: ", params["num_action"])
    params.update(dict(num_group=1 if args.actor_only else 2, T=args.T))
    return GCWrapper(GC, co, descriptors(args), gpu=args.gpu, use_numpy=False, params=params,
                     zero_planes=getattr(args, "preset_zero_planes", False))
