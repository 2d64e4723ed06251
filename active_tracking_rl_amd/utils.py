"""utils.py mirror: the pieces of the reference's utils.py the Track2D path uses (utils.py:11-21,92-95).
Weight initialisers live in model.py; ensure_shared_grads is replaced by Agent.allreduce_grads."""
import logging
import os


def setup_logger(logger_name, log_file, level=logging.INFO):
    l = logging.getLogger(logger_name)
    if l.handlers:          # the evaluator is called once per round here (a forked process in the reference)
        return
    formatter = logging.Formatter('%(asctime)s : %(message)s')
    fileHandler = logging.FileHandler(log_file, mode='w')
    fileHandler.setFormatter(formatter)
    streamHandler = logging.StreamHandler()
    streamHandler.setFormatter(formatter)
    l.setLevel(level)
    l.addHandler(fileHandler)
    l.addHandler(streamHandler)


def check_path(path):
    if not os.path.exists(path):
        os.makedirs(path, exist_ok=True)
