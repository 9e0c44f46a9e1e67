"""Importable alias of the hyphenated package directory `review-based-recommender_amd/`.

Python cannot import a directory whose name contains '-', so this shim points the
package `__path__` at the real directory and executes its `__init__.py` in place:
`import review_based_recommender_amd.functional` resolves to
`review-based-recommender_amd/functional.py`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "review-based-recommender_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
