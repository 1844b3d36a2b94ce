from .trainer import SimpleTrainer, default_argument_parser, launch_info

__all__ = ["SimpleTrainer", "default_argument_parser", "launch_info"]
