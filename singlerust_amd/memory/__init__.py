"""``single_rust::memory`` (src/memory/mod.rs:1-4): in-memory IMAnnData front-end."""
from . import processing, statistics

__all__ = ["processing", "statistics"]
