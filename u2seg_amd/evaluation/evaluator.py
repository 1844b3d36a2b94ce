"""Evaluator protocol and the inference loop (detectron2/evaluation/evaluator.py:17-224)."""
import torch
import torch.distributed as dist


def gather_to_rank0(obj):
    """[obj of rank 0, obj of rank 1, ...] on rank 0 and None elsewhere (utils/comm.py:167-218, comm.gather); a plain
    one-element list without a process group.  The test loader shards the images over the ranks (InferenceSampler), the
    evaluators meet again here."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


class DatasetEvaluator:
    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        pass


class DatasetEvaluators(DatasetEvaluator):
    """Several evaluators fed with the same inputs / outputs; their result dicts are merged (keys must not collide)."""

    def __init__(self, evaluators):
        self._evaluators = list(evaluators)

    def reset(self):
        for e in self._evaluators:
            e.reset()

    def process(self, inputs, outputs):
        for e in self._evaluators:
            e.process(inputs, outputs)

    def evaluate(self):
        results = {}
        for e in self._evaluators:
            r = e.evaluate()
            for k, v in (r or {}).items():
                assert k not in results, "Different evaluators produce results with the same key {}".format(k)
                results[k] = v
        return results


def inference_on_dataset(model, data_loader, evaluator):
    """model in eval mode over every batch of the loader, outputs handed to the evaluator; the model's previous
    training flag is restored afterwards."""
    if isinstance(evaluator, (list, tuple)):
        evaluator = DatasetEvaluators(evaluator)
    evaluator.reset()
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            for inputs in data_loader:
                evaluator.process(inputs, model(inputs))
    finally:
        model.train(was_training)
    return evaluator.evaluate() or {}
