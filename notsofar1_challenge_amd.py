"""Import alias: ``import notsofar1_challenge_amd`` -> the package directory ``notsofar1-challenge_amd/``.

The package directory carries the repository's name (with a hyphen, which Python's ``import``
statement cannot spell); this one-line shim makes it importable under a valid identifier.
"""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("notsofar1-challenge_amd")
