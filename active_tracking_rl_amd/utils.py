"""utils.py mirror: the pieces of the reference's utils.py the Track2D path uses (utils.py:11-21,92-95).
Weight initialisers live in model.py; ensure_shared_grads is replaced by Agent.allreduce_grads."""
import logging
import os

_LOG_FORMAT = '%(asctime)s : %(message)s'


def setup_logger(logger_name, log_file, level=logging.INFO):
    """Named logger writing to `log_file` (truncated) and to stderr, one '<time> : <message>' line per record
    (utils.py:11-21). Idempotent: the evaluator calls this once per round here (a forked process in the reference)."""
    log = logging.getLogger(logger_name)
    if log.handlers:
        return log
    log.setLevel(level)
    for handler in (logging.FileHandler(log_file, mode='w'), logging.StreamHandler()):
        handler.setFormatter(logging.Formatter(_LOG_FORMAT))
        log.addHandler(handler)
    return log


def check_path(path):
    os.makedirs(path, exist_ok=True)
