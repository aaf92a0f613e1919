"""Test-infrastructure stub for the `colorama` package (absent from this image).

bayes_opt imports it only for coloured log output (reference bayes_opt/target_space.py:10,
bayes_opt/logger.py:8); the stub maps every colour code to the empty string.
"""


class _Codes:
    def __getattr__(self, name):
        return ""


Fore = _Codes()
Back = _Codes()
Style = _Codes()


def just_fix_windows_console():
    return None


def init(*args, **kwargs):
    return None
