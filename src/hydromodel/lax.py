"""The explicit Lax scheme is broken in the reference snapshot and outside the
accelerated path; importing it works so that case scripts which merely import it keep loading."""


class LaxSolver:
    def __init__(self, *a, **k):
        raise NotImplementedError("LaxSolver is not part of the MI355X path (broken upstream, SURVEY.md F7)")
