from .line_search import line_search, limit_step_size

__all__ = ["line_search", "limit_step_size"]
