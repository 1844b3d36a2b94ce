from .config import CfgNode, configurable, get_cfg

__all__ = ["CfgNode", "configurable", "get_cfg"]
