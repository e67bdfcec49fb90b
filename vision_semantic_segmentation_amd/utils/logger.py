"""Minimal counterpart of src/utils/logger.py:11-91: stdout (+ optional version_N/log.txt)."""
import logging
import os
import sys


class MyLogger(object):
    def __init__(self, name, save_dir=None, use_timestamp=False, quiet=False):
        self.logger = logging.getLogger(name)
        self.logger.setLevel(logging.INFO if not quiet else logging.WARNING)
        self.logger.propagate = False
        if not self.logger.handlers:
            h = logging.StreamHandler(sys.stdout)
            h.setFormatter(logging.Formatter("[%(name)s] %(message)s"))
            self.logger.addHandler(h)
        self.save_dir = None
        if save_dir:
            version = 0
            while os.path.exists(os.path.join(save_dir, "version_%d" % version)):
                version += 1
            self.save_dir = os.path.join(save_dir, "version_%d" % version)
            os.makedirs(self.save_dir)
            fh = logging.FileHandler(os.path.join(self.save_dir, "log.txt"))
            self.logger.addHandler(fh)

    def log(self, msg):
        self.logger.info(msg)

    def warning(self, msg):
        self.logger.warning(msg)
