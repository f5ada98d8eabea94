"""Drop-in import surface: ``from label_anything.models import LabelAnything`` resolves to the MI355X
implementation in labelanything_amd (same constructor, state-dict layout and batch dictionary as the
reference package of the same name)."""
