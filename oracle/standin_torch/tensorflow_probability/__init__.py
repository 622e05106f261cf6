"""TEST INFRASTRUCTURE, CONTAINER-ONLY. Empty placeholder: the reference imports tensorflow_probability but the
DP-GP-LVM objective path never calls into it (see oracle/standin/tensorflow/__init__.py)."""
distributions = None
