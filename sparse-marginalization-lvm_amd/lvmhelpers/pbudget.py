"""PFactorBudget (lvmhelpers/pbudget.pyx:10-22): the Bernoulli factor with at most `budget` bits on
(lvmhelpers/budget.h:24-85)."""
from .pbernoulli import PFactorBernoulli


class PFactorBudget(PFactorBernoulli):

    def initialize(self, length, budget):
        self.n = int(length)
        self._budget = int(budget)
