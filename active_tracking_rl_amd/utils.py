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


class ScalarWriter(object):
    """`writer.add_scalar(tag, value, step)` of the reference's tensorboardX.SummaryWriter (train.py:17,98-104; test.py:19,
    94-98). tensorboardX when it is installed (requirements.txt:4 of the reference pins it; this image has no network),
    and always a JSON-lines file `<log_dir>/scalars.jsonl` with one {"tag", "value", "step"} record per call, so the same
    scalar names exist either way."""

    def __init__(self, log_dir):
        import json
        check_path(log_dir)
        self._json = json
        self._f = open(os.path.join(log_dir, "scalars.jsonl"), "a")
        self._tb = None
        try:
            from tensorboardX import SummaryWriter
            self._tb = SummaryWriter(log_dir)
        except Exception:
            pass

    def add_scalar(self, tag, value, step):
        value = float(value)
        self._f.write(self._json.dumps({"tag": tag, "value": value, "step": int(step)}) + "\n")
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()
        if self._tb is not None:
            self._tb.close()


def log_train_scalars(writer, stats, mode, fps, n_steps, num_agents=2):
    """The scalars train.py:97-104 of the reference writes per worker iteration, under the same names: stats = (policy_loss
    [A,1], value_loss [A,1], entropies [A,1], pred_loss [1,1]) as Agent.optimize returns them (means over envs here)."""
    policy_loss, value_loss, entropies, pred_loss = stats
    for i in range(min(num_agents, 3)):
        writer.add_scalar('train/policy_loss_' + str(i), policy_loss[i].mean().item(), n_steps)
        writer.add_scalar('train/value_loss_' + str(i), value_loss[i].mean().item(), n_steps)
        writer.add_scalar('train/entropies' + str(i), entropies[i].mean().item(), n_steps)
    writer.add_scalar('train/pred_R_loss', pred_loss.mean().item(), n_steps)
    writer.add_scalar('train/mode', mode, n_steps)
    writer.add_scalar('train/fps', fps, n_steps)
