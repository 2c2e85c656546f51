"""Device runtime: flat parameter arenas, graph capture, streams."""

from .arena import ArenaLayout, ArenaPool, ArenaStateDict, ParameterArena, Segment

__all__ = ["ArenaLayout", "ArenaPool", "ArenaStateDict", "ParameterArena", "Segment"]
