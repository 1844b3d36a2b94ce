"""Evaluator protocol and the inference loop (detectron2/evaluation/evaluator.py:17-224)."""
import torch


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
